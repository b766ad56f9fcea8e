# GPU session r02by: round-0 key width on the text workload under the quadrupled step
O=gpurun_out/r02by; mkdir -p $O
for kb in 64 56 48 40; do
BWTS_KEY_BITS=$kb timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --inverse-steps 1 --breakdown-steps 0 --no-cpu-baseline --no-e2e > $O/bench_$kb.log 2>&1; python3 -c "
import json
d=json.loads(open('$O/bench_$kb.log').read().strip().splitlines()[-1])
print('key bits $kb: forward ms', d['ms_per_step'], 'rounds', d['forward']['rounds'], 'tied', d['forward']['active_after_round0'], d['roundtrip_exact'])"
done
exit 0
