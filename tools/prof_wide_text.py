"""For rocprofv3 (GPU box): one forward of text(6 GiB) -- the n > 2^32 path with most positions tied.   python tools/prof_wide_text.py [GiB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
g = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = g << 30
pkg = ge.load_package(); ctx = pkg.Context(0)
a, b = ctx.alloc(n), ctx.alloc(n)
ctx.generate("text", 1, n, a)
t0 = time.perf_counter(); ctx.forward_device(a.ptr, n, b.ptr); dt = time.perf_counter() - t0
t = ctx.timings()
print("text(%d GiB) forward %.2f s = %.2f GB/s, rounds %d, tied %d" % (g, dt, n / 1e9 / dt, t.rounds, t.active_after_round0))
