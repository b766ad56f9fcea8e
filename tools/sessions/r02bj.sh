# GPU session r02bj: walk with 64-byte vs 16-byte symbol stores on the same box (inverse of zipf 2^30, alternating)
O=gpurun_out/r02bj; mkdir -p $O
for rep in 1 2; do
for syms in 64 16; do
BWTS_WALK_SYMS=$syms timeout -k 10 300 python bench.py --steps 3 --warmup 1 --inverse-steps 6 --no-cpu-baseline --no-e2e --no-text > $O/bench_$syms.json 2> $O/bench.err; python3 -c "
import json
d=json.loads(open('$O/bench_$syms.json').read().strip().splitlines()[-1])
print('syms $syms: inv', d['inverse_ms_per_step'], 'walk', d['inverse']['walk_ms_timed_region'], 'fwd', d['ms_per_step'], 'scatter', d['roofline']['ms_per_launch'], d['roundtrip_exact'])"
done; done
exit 0
