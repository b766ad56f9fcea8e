# GPU session r03bn: timing experiment -- what the random previous-symbol read of finishing elements costs the round kernel (a library variant that skips it: wrong bytes, timing only)
O=gpurun_out/r03bn; mkdir -p $O
for v in head nopv head nopv; do
  lib=""; [ $v = nopv ] && lib="$PWD/tools/ab/libbwts_nopv.so"
  BWTS_LIB_OVERRIDE=$lib timeout -k 10 300 python - $v <<'PY'
import sys, time
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
n = 1 << 30
a, b = ctx.alloc(n), ctx.alloc(n)
ctx.generate("text", 1, n, a)
ctx.forward_device(a, n, b)
ctx.set_timing(2)
ts = []
for r in range(3):
    t0 = time.perf_counter(); ctx.forward_device(a, n, b); ts.append(time.perf_counter() - t0)
k = ctx.timings().as_dict()["kernels"]
print(sys.argv[1], "forward ms", [round(1e3 * t, 1) for t in ts], "round class ms", round(k["round"]["ms"], 2))
PY
done
