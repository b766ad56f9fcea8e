# GPU session r03ag: run-to-run spread of the inverse walk over fresh processes (placement of the LF table), with two arena roundings
O=gpurun_out/r03ag; mkdir -p $O
cat > /tmp/inv_rep.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
n = 1 << 30
a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
ctx.generate("zipf", 1, n, a)
if len(sys.argv) > 2: ctx.inverse_device(a, n, c)      # inverse first: its arena is the context's first allocation
ctx.forward_device(a, n, b); ctx.inverse_device(b, n, c)
ctx.set_timing(1); ts = []
for r in range(3):
    t0 = time.perf_counter(); ctx.inverse_device(b, n, c); ts.append(1e3 * (time.perf_counter() - t0))
k = ctx.timings().as_dict()["kernels"]
print(sys.argv[1], "inverse ms", [round(t, 2) for t in ts], "walk", round(k["walk"]["ms"], 2), flush=True)
PY
for i in 1 2 3 4 5; do timeout -k 10 120 python /tmp/inv_rep.py default 2>&1 | tee -a $O/rep.txt; done
for i in 1 2 3 4 5; do BWTS_TEST_KNOBS=1 BWTS_ARENA_ALIGN_LOG2=30 timeout -k 10 120 python /tmp/inv_rep.py align1G 2>&1 | tee -a $O/rep.txt; done
for i in 1 2 3; do timeout -k 10 120 python /tmp/inv_rep.py invfirst x 2>&1 | tee -a $O/rep.txt; done
