# GPU session r02aw: counting loop of dense_round_kernel with four keys in flight -- the three texts
O=gpurun_out/r02aw; mkdir -p $O
timeout -k 10 300 python tools/time_realtext.py 26 5 > $O/realtext.log 2>&1; echo "rc=$?"; head -1 $O/realtext.log | cut -c1-200
timeout -k 10 300 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "rc=$?"; tail -2 $O/versions.log | head -1 | cut -c1-110
timeout -k 10 400 python bench.py --workload text --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_text.log 2>&1; echo "rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/bench_text.log').read().strip().splitlines()[-1])
print('text 2^30: forward ms', d.get('ms_per_step'), 'rounds', d.get('forward',{}).get('rounds'), d.get('roundtrip_exact'))"
exit 0
