# GPU session r03bk: LDS counting cap 128 / 192 / 256 now that larger groups go to WIDE chunks (A/B libraries built with -DDG_CAP=...)
O=gpurun_out/r03bk; mkdir -p $O
for v in 256 128 192 256 128 192; do
  lib=""; [ $v != 256 ] && lib="$PWD/tools/ab/libbwts_cap$v.so"
  echo "== cap $v"
  BWTS_LIB_OVERRIDE=$lib timeout -k 10 300 python tools/time_realtext.py 26 3 2>&1 | head -1
  BWTS_LIB_OVERRIDE=$lib timeout -k 10 300 python bench.py --workload text --no-e2e --no-cpu-baseline --steps 3 --inverse-steps 1 > $O/bench_text_$v.json 2> $O/bench_text_$v.err || { tail -3 $O/bench_text_$v.err; continue; }
  python - $v <<'PY'
import json, sys
v=sys.argv[1]
d=json.loads(open("gpurun_out/r03bk/bench_text_%s.json"%v).read().strip().splitlines()[-1])
print(v, "text fwd ms", d["ms_per_step"], d["roundtrip_exact"], {k:round(x["ms_per_launch"]*x["launches"]/3,2) for k,x in d["forward"]["kernels"].items() if k in ("round","rerank","radix_scatter","radix_hist")})
PY
done
