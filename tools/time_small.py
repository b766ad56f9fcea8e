"""Manual timing (GPU box): small and medium inputs resident in HBM -- wall time per forward / inverse call (timing off), launches per call
from the level-2 spans: where launch and host-sync latency, not bandwidth, sets the time.
    python tools/time_small.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
for kind in ("zipf", "text"):
    for log2n in (12, 16, 20, 24, 27):
        n = 1 << log2n
        a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
        ctx.generate(kind, 1, n, a)
        ctx.forward_device(a, n, b); ctx.inverse_device(b, n, c)
        tf, ti = [], []
        for r in range(5):
            t0 = time.perf_counter(); ctx.forward_device(a, n, b); tf.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); ctx.inverse_device(b, n, c); ti.append(time.perf_counter() - t0)
        t = ctx.timings()
        assert ctx.device_equal(a, c, n)
        print("%-5s 2^%-2d forward %8.3f ms = %7.1f MB/s   inverse %8.3f ms = %7.1f MB/s   rounds %d" % (
            kind, log2n, 1e3 * min(tf), n / 1e6 / min(tf), 1e3 * min(ti), n / 1e6 / min(ti), t.rounds), flush=True)
        for d in (a, b, c): d.free()
