# GPU session r02ag: experiment -- groups of up to DG_CAP elements ordered in LDS (library built with the DG_CAP of the caller)
O=gpurun_out/r02ag_$1; mkdir -p $O
timeout -k 10 300 python tools/time_realtext.py 26 5 > $O/realtext.log 2>&1; echo "rc=$?"; cat $O/realtext.log
timeout -k 10 300 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "rc=$?"; tail -3 $O/versions.log
timeout -k 10 400 python bench.py --workload text --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_text.log 2>&1; echo "rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/bench_text.log').read().strip().splitlines()[-1])
print('text 2^30: forward ms', d.get('ms_per_step'), 'rounds', d.get('forward',{}).get('rounds'))"
exit 0
