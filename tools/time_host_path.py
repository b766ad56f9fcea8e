"""Manual check (GPU box): the host-buffer entry points (bwts_forward / bwts_inverse) on a 1 GiB input -- wall time
including the pinned-staged H2D / D2H copies, next to the device time.  `value` in bench.py never includes these."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << log2n
x = O.generate("zipf", n, 1)
for name, fn in (("forward", ctx.forward), ("inverse", ctx.inverse)):
    for rep in range(3):
        t0 = time.perf_counter(); y = fn(x); dt = time.perf_counter() - t0
    t = ctx.timings()
    print("%s n=2^%d: wall %.1f ms = %.2f GB/s  (device %.1f ms, h2d %.1f ms, d2h %.1f ms)" % (
        name, log2n, 1e3 * dt, n / 1e9 / dt, t.total_ms, t.h2d_ms, t.d2h_ms))
    x = y if name == "forward" else x
