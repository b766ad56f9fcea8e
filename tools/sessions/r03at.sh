# GPU session r03at: groups of 257 .. 2048 members sorted as tiles of their own by the chunk kernel (LDS bitonic): parity, stress, then A/B against the cap-256 library
O=gpurun_out/r03at; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small or mid_size or deep_repeats or dense_ties or dense_rounds or text_16MiB or real_text or chunk_rounds or structured or known" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
STRESS_BUDGET_S=200 timeout -k 10 400 python tools/stress_chunks.py 60 0 > $O/stress.txt 2>&1; rc=$?; tail -3 $O/stress.txt
[ $rc -eq 0 ] || exit 1
for v in prev head prev head; do
  lib=""; [ $v = prev ] && lib="$PWD/tools/ab/libbwts_cap256.so"
  echo "== $v"
  BWTS_LIB_OVERRIDE=$lib timeout -k 10 300 python tools/time_realtext.py 26 3 2>&1 | grep -v "^\[chunks\]" | head -3
  BWTS_LIB_OVERRIDE=$lib timeout -k 10 300 python bench.py --workload text --no-e2e --no-cpu-baseline --steps 3 --inverse-steps 1 > $O/bench_text_$v.json 2> $O/bench_text_$v.err || { tail -3 $O/bench_text_$v.err; continue; }
  python - $v <<'PY'
import json, sys
v=sys.argv[1]
d=json.loads(open("gpurun_out/r03at/bench_text_%s.json"%v).read().strip().splitlines()[-1])
print(v, "text fwd ms", d["ms_per_step"], d["roundtrip_exact"], {k:round(x["ms_per_launch"]*x["launches"]/3,2) for k,x in d["forward"]["kernels"].items()} if "forward" in d and "kernels" in d["forward"] else "")
PY
done
