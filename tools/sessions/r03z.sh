# GPU session r03z: LDS group cap 256 (product) against 512: real text and text 2^30
O=gpurun_out/r03z; mkdir -p $O
run() { tag=$1
  timeout -k 10 200 python tools/time_realtext.py > $O/realtext_$tag.txt 2>&1; head -2 $O/realtext_$tag.txt | cut -c1-220
  timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --breakdown-steps 1 --inverse-steps 1 > $O/bench_$tag.json 2> $O/bench_$tag.err || { tail -3 $O/bench_$tag.err; return; }
  python - $tag <<'PY'
import json, sys
tag=sys.argv[1]
d=json.loads(open("gpurun_out/r03z/bench_%s.json"%tag).read().strip().splitlines()[-1])
print(tag, "text2^30", d["ms_per_step"], d["roundtrip_exact"], {k:round(v["ms_per_launch"]*v["launches"],1) for k,v in d["forward"]["kernels"].items()}, d.get("roofline"))
PY
}
run cap256
make -C bijective-bwt_amd HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -DDG_CAP=512" -j8 all > $O/build512.log 2>&1 || { tail $O/build512.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or text_16MiB or deep_repeats or real_text" > $O/pytest512.log 2>&1; tail -2 $O/pytest512.log
run cap512
