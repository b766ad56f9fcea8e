# GPU session r02d: spread counters, position-ordered groups, output prefault, D2H kernel/DMA split
set -o pipefail
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_text -- python3 $R/bench.py --workload text --no-e2e --no-cpu-baseline --steps 2 --warmup 1 --breakdown-steps 1 --inverse-steps 1 > $R/$O/bench_text.json 2> $R/$O/bench_text.err; echo "prof rc=$?")
BWTS_DENSE_ORDER=0 timeout -k 10 200 python bench.py --workload text --no-e2e --no-cpu-baseline --steps 2 --warmup 1 --breakdown-steps 1 --inverse-steps 1 > $O/bench_text_noorder.json 2> $O/bench_text_noorder.err; echo "noorder rc=$?"
for sp in 100 70 50 0; do BWTS_D2H_SPLIT=$sp timeout -k 10 120 python - 2>&1 <<'PY' | sed "s/^/split $sp: /"
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
n = 1 << 30
d = ctx.alloc(n); ctx.generate("zipf", 1, n, d); x = d.download(); d.free()
ctx.forward_into(x, np.empty(n, dtype=np.uint8))
for rep in range(3):
    out = np.empty(n, dtype=np.uint8)
    t0 = time.perf_counter(); ctx.forward_into(x, out); dt = time.perf_counter() - t0
    t = ctx.timings()
    print("wall %.1f ms  h2d %.1f device %.1f d2h %.1f" % (1e3 * dt, t.h2d_ms, t.total_ms, t.d2h_ms))
PY
done > $O/host_split.log; cat $O/host_split.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
