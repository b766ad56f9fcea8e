"""Manual (GPU box): where the inverse of the 1 GiB real text spends its wall time: timing fields at levels 0, 1, 2."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import realtext
x = np.frombuffer(realtext.corpus_big(1 << 30), dtype=np.uint8)
n = len(x)
pkg = ge.load_package(); ctx = pkg.Context(0)
a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
a.upload(x)
ctx.forward_device(a.ptr, n, b.ptr)
for lvl in (0,):
    ctx.set_timing(lvl)
    t0 = time.perf_counter(); ctx.inverse_device(b.ptr, n, c.ptr); w = time.perf_counter() - t0
    d = ctx.timings().as_dict()
    print("level", lvl, "wall ms %.2f" % (1e3 * w), json.dumps({k: v for k, v in d.items() if k != "round_active"}), flush=True)
