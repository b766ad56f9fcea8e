# GPU session r03r: chunk kernel with factor directory (FSL) / general instantiation, 40-bit key policy: parity subset (+ general Lyndon child), text timings
O=gpurun_out/r03r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or text_16MiB or structured or deep_repeats or mid_size or small or threshold or (alternate and (LYNDON or DENSE_STEP or KEY_SYMBOLS))" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > $O/bench_text.json 2> $O/bench_text.err || { tail -5 $O/bench_text.err; exit 1; }
timeout -k 10 200 python tools/time_realtext.py > $O/realtext.txt 2>&1; cat $O/realtext.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03r/bench_text.json").read().strip().splitlines()[-1])
print("text", d["ms_per_step"], d["roundtrip_exact"], d["forward"]["rounds"], d["forward"]["key_bits"], {k:round(v["ms_per_launch"]*v["launches"]/2,1) for k,v in d["forward"]["kernels"].items()})
PY
