# GPU session r02b: new inverse (two-level ranking, device-side tiny cycles), dense group-local rounds, PCIe copy kernel
set -o pipefail
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 120 python tools/time_inverse.py zipf 30 9 8 7 6 5 > $O/inverse_g.log 2>&1; echo "inv rc=$?"; cat $O/inverse_g.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for m in kernel dma; do BWTS_D2H=$m BWTS_H2D=$m timeout -k 10 120 python - > $O/host_$m.log 2>&1 <<'PY'
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
echo "host $m rc=$?"; cat $O/host_$m.log; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_text -- python3 $R/bench.py --workload text --no-e2e --no-cpu-baseline --steps 2 --warmup 1 --breakdown-steps 0 --inverse-steps 1 > $R/$O/bench_text.json 2> $R/$O/bench_text.err; echo "prof rc=$?"
