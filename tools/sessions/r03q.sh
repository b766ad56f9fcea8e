# GPU session r03q: text workloads under forced round-0 key widths, then the chunk kernel with 256-thread workgroups
O=gpurun_out/r03q; mkdir -p $O
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --breakdown-steps 1 --inverse-steps 1 > $O/bench_$tag.json 2> $O/bench_$tag.err || { tail -3 $O/bench_$tag.err; return; }
  env "$@" timeout -k 10 200 python tools/time_realtext.py > $O/realtext_$tag.txt 2>&1
  python - $tag <<'PY'
import json, sys
tag=sys.argv[1]
d=json.loads(open("gpurun_out/r03q/bench_%s.json"%tag).read().strip().splitlines()[-1])
print(tag, "text2^30", d["ms_per_step"], d["roundtrip_exact"], "rounds", d["forward"]["rounds"], "kb", d["forward"]["key_bits"], "active", d["forward"]["round_active"][:3], {k:round(v["ms_per_launch"]*v["launches"],1) for k,v in d["forward"]["kernels"].items()})
print("   ", open("gpurun_out/r03q/realtext_%s.txt"%tag).read().splitlines()[0][:160])
PY
}
run kb64
run kb48 BWTS_KEY_BITS=48
run kb40 BWTS_KEY_BITS=40
run kb32 BWTS_KEY_BITS=32
make -C bijective-bwt_amd HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -DCH_THREADS=256 -DCH_FS=128" -j8 all > $O/build256.log 2>&1 || { tail $O/build256.log; exit 1; }
run t256_kb64
run t256_kb40 BWTS_KEY_BITS=40
