# GPU session r03ai: chunk kernel shape 512 threads x 4 slots (product) against 256 x 8 (same tile, same LDS, half the waves, twice the loads per lane)
O=gpurun_out/r03ai; mkdir -p $O
run() { tag=$1
  timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --breakdown-steps 1 --inverse-steps 1 > $O/bench_$tag.json 2> $O/bench_$tag.err || { tail -3 $O/bench_$tag.err; return; }
  timeout -k 10 200 python tools/time_realtext.py > $O/realtext_$tag.txt 2>&1
  python - $tag <<'PY'
import json, sys
tag=sys.argv[1]
d=json.loads(open("gpurun_out/r03ai/bench_%s.json"%tag).read().strip().splitlines()[-1])
print(tag, "text2^30", d["ms_per_step"], d["roundtrip_exact"], "round", d["roofline"]["kernel"][:5], {k:round(v["ms_per_launch"]*v["launches"],1) for k,v in d["forward"]["kernels"].items() if k in ("round","rerank")}, open("gpurun_out/r03ai/realtext_%s.txt"%tag).read().splitlines()[0][50:110])
PY
}
run p512x4
make -C bijective-bwt_amd HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -DCH_THREADS=256 -DCH_ITEMS=8 -DCH_MIN_WAVES=4" -j8 all > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or text_16MiB or deep_repeats or real_text or chunk_rounds" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
run p256x8
run p256x8
