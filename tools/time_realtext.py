"""Manual timing (GPU box): the real-text corpus of check_realtext.py, resident in HBM, forward transform untimed inside (timing level 0):
wall time per call next to the sum of the device spans of one extra call at level 2 -- the difference is launch and host-sync time.
    python tools/time_realtext.py [log2limit] [reps]"""
import os, sys, time, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
src = open(os.path.join(ROOT, "tools", "check_realtext.py")).read().split("limit = (1 <<")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "check_realtext.py")}
exec(src, ns)
limit = (1 << int(sys.argv[1])) if len(sys.argv) > 1 else 1 << 26
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x = np.frombuffer(ns["corpus"](limit), dtype=np.uint8)
n = len(x)
pkg = ge.load_package(); ctx = pkg.Context(0)
a, b = ctx.alloc(n), ctx.alloc(n)
a.upload(x)
ctx.forward_device(a.ptr, n, b.ptr)
ts = []
for r in range(reps):
    t0 = time.perf_counter(); ctx.forward_device(a.ptr, n, b.ptr); ts.append(time.perf_counter() - t0)
t = ctx.timings()
print("real text n=%d (%.1f MiB): forward wall ms %s  best %.2f = %.2f GB/s   rounds %d tied %d" % (
    n, n / 2**20, [round(1e3 * v, 2) for v in ts], 1e3 * min(ts), n / 1e9 / min(ts), t.rounds, t.active_after_round0))
ctx.set_timing(2)
ctx.forward_device(a.ptr, n, b.ptr)
k = ctx.timings().as_dict()
dev = sum(v["ms"] for v in k["kernels"].values())
print("level-2 spans: %.2f ms in kernels (%d launches)" % (dev, sum(v["launches"] for v in k["kernels"].values())), {c: round(v["ms"], 2) for c, v in k["kernels"].items()})
print("round_active", k.get("round_active"))
