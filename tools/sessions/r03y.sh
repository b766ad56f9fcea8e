# GPU session r03y: inverse walk with the next LF entry prefetched and one atomic for count + sum: parity, then A/B against the previous commit's library
O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inverse or small or mid_size or structured or known or reference_unbwts or stray or batch" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python tools/diag_wide_narrow.py 2>&1 | tee $O/wide_narrow.txt
for v in prev head prev head; do
  lib=""; [ $v = prev ] && lib="$PWD/tools/ab/libbwts_prev.so"
  BWTS_LIB_OVERRIDE=$lib timeout -k 10 300 python bench.py --no-text --no-e2e --no-cpu-baseline --steps 3 --inverse-steps 5 > $O/bench_$v.json 2> $O/bench_$v.err || { tail -3 $O/bench_$v.err; continue; }
  python - $v <<'PY'
import json, sys
v=sys.argv[1]
d=json.loads(open("gpurun_out/r03y/bench_%s.json"%v).read().strip().splitlines()[-1])
print(v, "fwd", d["ms_per_step"], "inv", d["inverse_ms_per_step"], d["roundtrip_exact"], "walk", d["inverse"]["walk_ms_timed_region"], {k:round(x["ms_per_launch"]*x["launches"]/2,2) for k,x in d["inverse"]["kernels"].items()})
PY
done
