# GPU session r02ai: fused column scan -- fuzz, texts, full suite, headline bench
O=gpurun_out/r02ai; mkdir -p $O
timeout -k 10 300 python tools/stress_random.py 200 4000 > $O/stress_default.log 2>&1; echo "default fuzz rc=$?"; tail -2 $O/stress_default.log
timeout -k 10 300 python tools/time_realtext.py 26 5 > $O/realtext.log 2>&1; echo "rc=$?"; head -2 $O/realtext.log | cut -c1-330
timeout -k 10 300 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "rc=$?"; tail -2 $O/versions.log | cut -c1-120
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "full suite rc=$?"; tail -3 $O/full.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('zipf fwd ms', d['ms_per_step'], 'inv ms', d['inverse_ms_per_step'], 'scatter ms', d['roofline']['ms_per_launch'], 'text', d['text']['forward_ms'], 'e2e', d['e2e']['host_forward_MBps'], d['e2e']['host_inverse_MBps'])"
exit 0
