# GPU session r02aj: quadrupled step in the group-local rounds -- fuzz, texts, full suite
O=gpurun_out/r02aj; mkdir -p $O
timeout -k 10 300 python tools/stress_random.py 200 6000 > $O/stress_default.log 2>&1; echo "default fuzz rc=$?"; tail -2 $O/stress_default.log
timeout -k 10 300 python tools/time_realtext.py 26 5 > $O/realtext.log 2>&1; echo "rc=$?"; head -3 $O/realtext.log | cut -c1-330
timeout -k 10 300 python tools/check_realtext.py 26 > $O/realtext_check.log 2>&1; echo "rc=$?"; tail -2 $O/realtext_check.log | cut -c1-160
timeout -k 10 300 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "rc=$?"; tail -2 $O/versions.log | cut -c1-120
timeout -k 10 400 python bench.py --workload text --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_text.log 2>&1; echo "rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/bench_text.log').read().strip().splitlines()[-1])
print('text 2^30: forward ms', d.get('ms_per_step'), 'rounds', d.get('forward',{}).get('rounds'), d.get('roundtrip_exact'), d.get('forward',{}).get('round_active'))"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "full suite rc=$?"; tail -3 $O/full.log
exit 0
