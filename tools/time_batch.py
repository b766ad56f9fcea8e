"""GPU box: bwts_forward_batch / bwts_inverse_batch over `items` zipf inputs of 2^log2n bytes (unpinned numpy buffers), next to the
single-call rate.  BWTS_BATCH_TRACE=1 prints how long each pipeline stage was busy.   python tools/time_batch.py [log2n] [items]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
items = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << log2n
pkg = ge.load_package(); ctx = pkg.Context(0)
tmp = ctx.alloc(n); xs = []
for sd in range(1, items + 1):
    ctx.generate("zipf", sd, n, tmp); xs.append(tmp.download())
tmp.free()
y0 = ctx.forward(xs[0]); ctx.forward_batch(xs[:2])
t0 = time.perf_counter(); y1 = ctx.forward(xs[0]); single = time.perf_counter() - t0
for rep in range(2):
    t0 = time.perf_counter(); ys = ctx.forward_batch(xs); bf = time.perf_counter() - t0
    t0 = time.perf_counter(); bs = ctx.inverse_batch(ys); bi = time.perf_counter() - t0
    print("batch %d x 2^%d: forward %.1f ms = %.1f GB/s, inverse %.1f ms = %.1f GB/s (single forward call %.1f ms = %.1f GB/s); exact %s" % (
        items, log2n, 1e3 * bf, items * n / 1e9 / bf, 1e3 * bi, items * n / 1e9 / bi, 1e3 * single, n / 1e9 / single,
        bool(np.array_equal(ys[0], y0) and all(np.array_equal(b, x) for b, x in zip(bs, xs)))), flush=True)
