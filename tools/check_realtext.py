"""Manual check (GPU box): real text from the image itself (source and documentation files under /usr/lib, /usr/share),
concatenated up to a size limit.  Forward / inverse timings, round structure, and -- below 64 MiB -- equality with the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge

from realtext import corpus          # tests/realtext.py: the same corpus the gpu test uses

limit = (1 << int(sys.argv[1])) if len(sys.argv) > 1 else 1 << 26
x = np.frombuffer(corpus(limit), dtype=np.uint8)
pkg = ge.load_package(); ctx = pkg.Context(0); ctx.set_timing(2)
for rep in range(2):
    y = ctx.forward(x)
tm = ctx.timings().as_dict()
print("real text n=%d (%.1f MiB) sigma=%d key_bits=%d rounds=%d active0=%d (%.1f%%) fwd device %.1f ms (%.0f MB/s) kernels %s" % (
    len(x), len(x) / 2**20, len(np.unique(x)), tm["key_bits"], tm["rounds"], tm["active_after_round0"], 100.0 * tm["active_after_round0"] / len(x),
    tm["total_ms"], len(x) / 1e3 / tm["total_ms"], {k: round(v["ms"], 1) for k, v in tm["kernels"].items()}))
back = ctx.inverse(y)
ti = ctx.timings().as_dict()
print("roundtrip", bool(np.array_equal(back, x)), "inv device %.1f ms (%.0f MB/s) cycles %d" % (ti["total_ms"], len(x) / 1e3 / ti["total_ms"], ti["factors"]))
if len(x) <= (1 << 26):
    t0 = time.time(); w = O.forward(x); print("oracle %.1f s equal=%s" % (time.time() - t0, bool(np.array_equal(w, y))))
