# GPU session r02ah: DG_CAP 128 / DG_ITEMS 4 as defaults -- fuzz, full suite, headline bench
O=gpurun_out/r02ah; mkdir -p $O
timeout -k 10 400 python tools/stress_random.py 200 3000 > $O/stress_default.log 2>&1; echo "default fuzz rc=$?"; tail -2 $O/stress_default.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "full suite rc=$?"; tail -3 $O/full.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('zipf fwd ms', d['ms_per_step'], 'inv ms', d['inverse_ms_per_step'], 'text', d['text']['forward_ms'], 'e2e', d['e2e']['host_forward_MBps'], d['e2e']['host_inverse_MBps'])"
exit 0
