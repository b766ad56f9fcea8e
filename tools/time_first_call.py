import sys, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package()
t0 = time.time(); ctx = pkg.Context(0); print("ctx create %.0f ms" % (1e3 * (time.time() - t0)))
for log2n in (24, 30):
    n = 1 << log2n
    a, b = ctx.alloc(n), ctx.alloc(n)
    ctx.generate("zipf", 1, n, a)
    for rep in range(3):
        t0 = time.time(); ctx.forward_device(a, n, b); print("n=2^%d forward call %d: wall %.1f ms (device %.1f ms)" % (log2n, rep, 1e3 * (time.time() - t0), ctx.timings().total_ms), flush=True)
    for rep in range(2):
        t0 = time.time(); ctx.inverse_device(b, n, a); print("n=2^%d inverse call %d: wall %.1f ms (device %.1f ms)" % (log2n, rep, 1e3 * (time.time() - t0), ctx.timings().total_ms), flush=True)
