import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get('BWTS_LIB'): pkg.LIB_PATH = os.environ['BWTS_LIB']
ctx = pkg.Context(0)
n = 1 << 30
a, b = ctx.alloc(n), ctx.alloc(n)
ctx.generate("zipf", 1, n, a)
for rep in range(2):
    try:
        ctx.inverse_device(a, n, b)
        k = ctx.timings().as_dict()
        print(os.environ.get("BWTS_SPLIT_LOG2"), os.environ.get("BWTS_EXP_NOMARK"), "total %.1f" % k["total_ms"], {x: round(v["ms"], 2) for x, v in k["kernels"].items()}, "unv", k["unvisited"])
    except Exception as e:
        print("fail", e)
