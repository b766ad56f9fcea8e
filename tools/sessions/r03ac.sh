# GPU session r03ac: list order by group records: parity subset, then A/B against the element sort (previous commit) on text 2^30 and real text
O=gpurun_out/r03ac; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or text_16MiB or structured or deep_repeats or mid_size or real_text or threshold or (alternate and (LYNDON or DENSE_STEP or KEY_SYMBOLS))" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for v in prev head prev head; do
  lib=""; [ $v = prev ] && lib="$PWD/tools/ab/libbwts_prev.so"
  BWTS_LIB_OVERRIDE=$lib timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --breakdown-steps 1 --inverse-steps 1 > $O/bench_$v.json 2> $O/bench_$v.err || { tail -3 $O/bench_$v.err; continue; }
  BWTS_LIB_OVERRIDE=$lib timeout -k 10 200 python tools/time_realtext.py > $O/realtext_$v.txt 2>&1
  python - $v <<'PY'
import json, sys
v=sys.argv[1]
d=json.loads(open("gpurun_out/r03ac/bench_%s.json"%v).read().strip().splitlines()[-1])
k=d["forward"]["kernels"]
print(v, "text2^30", d["ms_per_step"], d["roundtrip_exact"], {n:round(x["ms_per_launch"]*x["launches"],1) for n,x in k.items() if n in ("round","rerank","radix_scatter","radix_hist")}, open("gpurun_out/r03ac/realtext_%s.txt"%v).read().splitlines()[0][50:120])
PY
done
