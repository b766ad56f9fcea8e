"""Manual (GPU box): group-size classes of the chunk lists, round by round (BWTS_TEST_KNOBS=1 BWTS_ROUND_TRACE=1), for text(2^30), the
1 GiB real text and the 53.6 MiB real text.    python tools/diag_group_sizes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import realtext
pkg = ge.load_package(); ctx = pkg.Context(0)
n = 1 << 30
a, b = ctx.alloc(n), ctx.alloc(n)
print("== text(2^30)", file=sys.stderr, flush=True)
ctx.generate("text", 1, n, a)
ctx.forward_device(a.ptr, n, b.ptr)
print("== real text 1 GiB", file=sys.stderr, flush=True)
a.upload(np.frombuffer(realtext.corpus_big(n), dtype=np.uint8))
ctx.forward_device(a.ptr, n, b.ptr)
x = np.frombuffer(realtext.corpus(1 << 26), dtype=np.uint8)
print("== real text %d bytes" % len(x), file=sys.stderr, flush=True)
c = ctx.alloc(len(x)); c.upload(x)
ctx.forward_device(c.ptr, len(x), b.ptr)
