# GPU session r03af: inverse walk with / without the LF prefetch, at 2^30 (zipf) and 2^32 (dna), alternating on one box
O=gpurun_out/r03af; mkdir -p $O
cat > /tmp/inv_ab.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
for kind, log2n in (("zipf", 30), ("dna", 32)):
    n = 1 << log2n
    a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    ctx.generate(kind, 1, n, a); ctx.forward_device(a, n, b); ctx.inverse_device(b, n, c)
    ctx.set_timing(1); ts = []
    for r in range(4):
        t0 = time.perf_counter(); ctx.inverse_device(b, n, c); ts.append(1e3 * (time.perf_counter() - t0))
    k = ctx.timings().as_dict()["kernels"]
    print(sys.argv[1], kind, log2n, "inverse ms", [round(t, 2) for t in ts], "walk", round(k["walk"]["ms"], 2), "exact", ctx.device_equal(a, c, n), flush=True)
    for x in (a, b, c): x.free()
PY
for v in nopre head nopre head; do
  lib=""; [ $v = nopre ] && lib="$PWD/tools/ab/libbwts_nopre.so"
  BWTS_LIB_OVERRIDE=$lib timeout -k 10 300 python /tmp/inv_ab.py $v 2>&1 | tee -a $O/ab.txt
done
