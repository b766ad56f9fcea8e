"""GPU box: one forward transform of text(2^28) (plus a warm-up), for rocprofv3 --pmc passes over the round kernels."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
a, b = ctx.alloc(n), ctx.alloc(n)
ctx.generate("text", 1, n, a)
for rep in range(2):
    ctx.forward_device(a, n, b)
print("total ms", ctx.timings().total_ms)
