# GPU session r02cb: position order on/off by input size (versions text 215 MiB, text 2^26 .. 2^28)
O=gpurun_out/r02cb; mkdir -p $O
for ord in 1 0; do
BWTS_DENSE_ORDER=$ord timeout -k 10 200 python tools/check_versions_text.py 2>&1 | grep "versions x4" | cut -c1-110
for l in 26 27 28; do
BWTS_DENSE_ORDER=$ord timeout -k 10 300 python bench.py --workload text --log2n $l --steps 3 --warmup 1 --inverse-steps 1 --breakdown-steps 0 --no-cpu-baseline --no-e2e --no-text > $O/b.log 2>&1; python3 -c "
import json
d=json.loads(open('$O/b.log').read().strip().splitlines()[-1])
print('order $ord: text 2^$l forward ms', d['ms_per_step'])"
done; done
exit 0
