"""Manual check (GPU box): the host-buffer entry points (bwts_forward / bwts_inverse) on a 2^log2n-byte input -- wall time
including the staged PCIe copies, next to the device time, for several copy-worker counts (BWTS_COPY_THREADS), for pinned
caller blocks (bwts_host_alloc) and through the sink callback.  `value` in bench.py never includes these.

    python tools/time_host_path.py [log2n] [kind]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
kind = sys.argv[2] if len(sys.argv) > 2 else "zipf"
n = 1 << log2n

def line(tag, dt, t):
    print("%-34s wall %7.1f ms = %6.2f GB/s  (h2d %6.1f  device %6.1f  d2h %6.1f)" % (tag, 1e3 * dt, n / 1e9 / dt, t.h2d_ms, t.total_ms, t.d2h_ms), flush=True)

x = None
for threads in [int(v) for v in os.environ.get("BWTS_SWEEP_THREADS", "6,1,2,4,8,12").split(",")]:
    os.environ["BWTS_COPY_THREADS"] = str(threads)
    ctx = pkg.Context(0)
    if x is None:
        d = ctx.alloc(n); ctx.generate(kind, 1, n, d); x = d.download(); d.free()
    ctx.forward_into(x, np.empty(n, dtype=np.uint8))            # warm: arena, staging, workers
    best = None
    for rep in range(3):
        out = np.empty(n, dtype=np.uint8)                        # fresh output: first-touch faults included
        t0 = time.perf_counter(); ctx.forward_into(x, out); dt = time.perf_counter() - t0
        if best is None or dt < best[0]: best = (dt, ctx.timings())
    line("forward, %2d copy threads, fresh out" % threads, *best)
    t0 = time.perf_counter(); ctx.forward_into(x, out); dt = time.perf_counter() - t0
    line("forward, %2d copy threads, reused out" % threads, dt, ctx.timings())
    y = out
    back = np.empty(n, dtype=np.uint8)
    t0 = time.perf_counter(); ctx.inverse_into(y, back); dt = time.perf_counter() - t0
    line("inverse, %2d copy threads, fresh out" % threads, dt, ctx.timings())
    assert np.array_equal(back, x)
    if threads == 6:
        t0 = time.perf_counter(); a, pa = ctx.host_alloc(n); b, pb = ctx.host_alloc(n); dt = time.perf_counter() - t0
        print("bwts_host_alloc of 2 x 2^%d bytes: %.1f ms" % (log2n, 1e3 * dt))
        a[:] = x
        for rep in range(2):
            t0 = time.perf_counter(); ctx.forward_into(a, b); dt = time.perf_counter() - t0
        line("forward, pinned in and out", dt, ctx.timings())
        assert np.array_equal(b, y)
        del a, b; ctx.host_free(pa); ctx.host_free(pb)
        sink_bytes = [0]
        def take(user, ptr, length):
            sink_bytes[0] += length
            return 0
        import ctypes
        cb = pkg.SINK_FN(take)
        t0 = time.perf_counter(); rc = pkg.lib().bwts_forward_sink(ctx._h, x.ctypes.data, n, cb, None); dt = time.perf_counter() - t0
        assert rc == 0 and sink_bytes[0] == n
        line("forward, sink that drops the data", dt, ctx.timings())
    ctx.close()
