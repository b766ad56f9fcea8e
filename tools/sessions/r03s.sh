# GPU session r03s: A/B on one box: chunk kernel variants (v00 first form, v0 tables + binary search, head = directory), text 2^30 with 40-bit keys
O=gpurun_out/r03s; mkdir -p $O
for rep in 1 2; do
for v in v00 v0 head; do
  lib=""; [ $v != head ] && lib="$PWD/tools/ab/libbwts_$v.so"
  BWTS_LIB_OVERRIDE=$lib BWTS_KEY_BITS=40 timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --breakdown-steps 1 --inverse-steps 1 > $O/bench_$v.json 2> $O/bench_$v.err || { tail -3 $O/bench_$v.err; continue; }
  python - $v <<'PY'
import json, sys
v=sys.argv[1]
d=json.loads(open("gpurun_out/r03s/bench_%s.json"%v).read().strip().splitlines()[-1])
k=d["forward"]["kernels"]
print(v, "text2^30 kb40", d["ms_per_step"], d["roundtrip_exact"], "round", round(k["round"]["ms_per_launch"]*k["round"]["launches"],1), "rerank", round(k["rerank"]["ms_per_launch"]*k["rerank"]["launches"],1))
PY
done
done
