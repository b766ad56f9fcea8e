# GPU session r03a: first light of the chunked dense rounds (chunk_rounds.h): parity subset, then text timings A/B against the tile form
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or text_16MiB or structured or deep_repeats or mid_size or small or threshold" > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
BWTS_ROUND_TRACE=1 timeout -k 10 300 python bench.py --workload text --steps 2 --warmup 1 --no-e2e --no-cpu-baseline > $O/bench_text_chunks.json 2> $O/bench_text_chunks.err || { tail -5 $O/bench_text_chunks.err; exit 1; }
BWTS_DENSE=tiles timeout -k 10 300 python bench.py --workload text --steps 2 --warmup 1 --no-e2e --no-cpu-baseline > $O/bench_text_tiles.json 2> $O/bench_text_tiles.err || { tail -5 $O/bench_text_tiles.err; exit 1; }
timeout -k 10 200 python tools/time_realtext.py > $O/realtext_chunks.txt 2>&1 || { tail -5 $O/realtext_chunks.txt; exit 1; }
BWTS_DENSE=tiles timeout -k 10 200 python tools/time_realtext.py > $O/realtext_tiles.txt 2>&1
python - <<'PY'
import json
for f in ("chunks","tiles"):
    d=json.loads(open("gpurun_out/r03a/bench_text_%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roundtrip_exact"], d["forward"]["rounds"], {k:(v["ms_per_launch"],v["launches"]) for k,v in d["forward"]["kernels"].items()})
PY
cat $O/realtext_chunks.txt $O/realtext_tiles.txt
