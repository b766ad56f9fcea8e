# GPU session r02ca: the position order of the tied list under the quadrupled step (text 2^30, real text)
O=gpurun_out/r02ca; mkdir -p $O
for ord in 1 0; do
BWTS_DENSE_ORDER=$ord timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --inverse-steps 1 --breakdown-steps 0 --no-cpu-baseline --no-e2e > $O/bench_$ord.log 2>&1; python3 -c "
import json
d=json.loads(open('$O/bench_$ord.log').read().strip().splitlines()[-1])
print('order $ord: text forward ms', d['ms_per_step'], 'rounds', d['forward']['rounds'], d['roundtrip_exact'])"
BWTS_DENSE_ORDER=$ord timeout -k 10 200 python tools/time_realtext.py 26 3 2>&1 | head -1 | cut -c1-150
done
exit 0
