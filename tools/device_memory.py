"""Manual check (GPU box): device memory a context holds per input byte after one inverse / one forward of zipf 2^log2n
(the caller's device-resident in/out buffers not counted).    python tools/device_memory.py [log2n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 30)
for which in ("inverse", "forward"):
    ctx = pkg.Context(0)
    a, b = ctx.alloc(n), ctx.alloc(n)
    ctx.generate("zipf", 1, n, a)
    (ctx.inverse_device if which == "inverse" else ctx.forward_device)(a, n, b)
    print(which, "device bytes / n = %.1f" % (ctx.timings().device_bytes / n))
    a.free(); b.free(); ctx.close()
