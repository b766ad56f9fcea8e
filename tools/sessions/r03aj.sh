# GPU session r03aj: expand from the suffix array: parity subset + text timings
O=gpurun_out/r03aj; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or text_16MiB or structured or deep_repeats or mid_size or real_text or chunk_rounds or threshold or (alternate and (LYNDON or DENSE_STEP or KEY_SYMBOLS))" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --breakdown-steps 1 --inverse-steps 1 > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
timeout -k 10 200 python tools/time_realtext.py > $O/realtext.txt 2>&1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03aj/bench.json").read().strip().splitlines()[-1])
print("text2^30", d["ms_per_step"], d["roundtrip_exact"], {k:round(v["ms_per_launch"]*v["launches"],1) for k,v in d["forward"]["kernels"].items() if k in ("round","rerank","radix_scatter","radix_hist")}, open("gpurun_out/r03aj/realtext.txt").read().splitlines()[0][50:110])
PY
done
timeout -k 10 300 python tools/check_text_2p32.py 32 2>&1 | tail -2
