# GPU session r02bk: inverse with per-range moments instead of the index log -- tests, fuzz, A/B bench on one box
O=gpurun_out/r02bk; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inverse or small or mid_size or tiny or long_cycle or low_entropy or reference_unbwts or golden or kat" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 300 python tools/stress_random.py 300 10000 > $O/stress.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/stress.log
for mode in moments log moments log; do
BWTS_INV_MARK=$mode timeout -k 10 300 python bench.py --steps 3 --warmup 1 --inverse-steps 6 --no-cpu-baseline --no-e2e --no-text > $O/bench_$mode.json 2> $O/bench.err; python3 -c "
import json
d=json.loads(open('$O/bench_$mode.json').read().strip().splitlines()[-1])
print('$mode: inv', d['inverse_ms_per_step'], 'walk', d['inverse']['walk_ms_timed_region'], {k:round(v['ms_per_launch']*v['launches']/2,2) for k,v in d['inverse']['kernels'].items()}, d['roundtrip_exact'], 'unvisited', d['inverse']['unvisited'])"
done
exit 0
