"""Manual check (GPU box): inverse timing per kernel class for several splitter spacings (BWTS_SPLIT_LOG2).
    python tools/time_inverse.py [kind] [log2n] [g ...]"""
import os as _os; _os.environ.setdefault("BWTS_TEST_KNOBS", "1")      # this tool drives alternate-path knobs
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 30)
gs = [int(v) for v in sys.argv[3:]] or [None]
ctx = pkg.Context(0)
ctx.set_timing(2)
a, b = ctx.alloc(n), ctx.alloc(n)
ctx.generate(kind, 1, n, a)
for g in gs:
    if g is None: os.environ.pop("BWTS_SPLIT_LOG2", None)
    else: os.environ["BWTS_SPLIT_LOG2"] = str(g)
    for rep in range(2):
        ctx.inverse_device(a, n, b)
        k = ctx.timings().as_dict()
    print("g", g, "total %.2f" % k["total_ms"], {x: round(v["ms"], 2) for x, v in k["kernels"].items()}, "unv", k["unvisited"], "cycles", k["factors"], flush=True)
