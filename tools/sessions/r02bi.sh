# GPU session r02bi: walk with 64-byte symbol stores -- inverse tests, fuzz, bench (inverse ms), 12 GiB 64-bit inverse
O=gpurun_out/r02bi; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inverse or small or mid_size or tiny or long_cycle or low_entropy or wide_path_small" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 300 python tools/stress_random.py 200 9500 > $O/stress.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/stress.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-text > $O/bench.json 2> $O/bench.err; python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('zipf fwd', d['ms_per_step'], 'inv', d['inverse_ms_per_step'], 'walk', d['inverse']['walk_ms_timed_region'], 'scatter', d['roofline']['ms_per_launch'], d['roundtrip_exact'])"
timeout -k 10 400 python tools/run_wide.py 12 > $O/wide12.log 2>&1; echo "wide rc=$?"; grep "inverse\|round trip" $O/wide12.log | cut -c1-260
exit 0
