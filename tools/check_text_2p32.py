"""Manual check (GPU box): text(2^32) -- most positions tied at the largest n the main path takes (dense rounds without the
position order, which needs 2 n < 2^32): device-side round trip, histogram, timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 32)
ctx = pkg.Context(0)
a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
ctx.generate("text", 1, n, a)
try:
    t0 = time.perf_counter(); ctx.forward_device(a, n, b); dt = time.perf_counter() - t0
except pkg.BwtsError as e:
    print("forward failed:", e); sys.exit(0)
t0 = time.perf_counter(); ctx.forward_device(a, n, b); dt2 = time.perf_counter() - t0
print("first call %.0f ms (arenas allocated), second call:" % (1e3 * dt), flush=True)
dt = dt2
k = ctx.timings().as_dict()
print("forward %.0f ms = %.2f GB/s rounds %d tied %d key_bits %d device GiB %.0f" % (1e3 * dt, n / 1e9 / dt, k["rounds"], k["active_after_round0"], k["key_bits"], k["device_bytes"] / 2**30), flush=True)
t0 = time.perf_counter(); ctx.inverse_device(b, n, c); dt = time.perf_counter() - t0
print("inverse %.0f ms = %.2f GB/s" % (1e3 * dt, n / 1e9 / dt), "round trip exact:", ctx.device_equal(a, c, n), flush=True)
