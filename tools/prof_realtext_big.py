"""For rocprofv3 (GPU box): three forwards of the 1 GiB real text, nothing else.   python tools/prof_realtext_big.py [forwards]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import realtext
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
x = np.frombuffer(realtext.corpus_big(1 << 30), dtype=np.uint8)
n = len(x)
pkg = ge.load_package(); ctx = pkg.Context(0)
a, b = ctx.alloc(n), ctx.alloc(n)
a.upload(x)
for r in range(reps):
    ctx.forward_device(a.ptr, n, b.ptr)
print("forwards", reps, "rounds", ctx.timings().rounds)
