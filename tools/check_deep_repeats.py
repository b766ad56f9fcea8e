"""Manual check (GPU box): inputs with long repeats -> many doubling rounds with a large tied set."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0); ctx.set_timing(2)
log2b = int(sys.argv[1]) if len(sys.argv) > 1 else 22
block = O.generate("zipf", 1 << log2b, 5)
x = np.concatenate([block, block, block[: len(block) // 2], O.generate("zipf", 1000, 6), block])
for rep in range(2):
    t0 = time.perf_counter(); y = ctx.forward(x); dt = time.perf_counter() - t0
tm = ctx.timings().as_dict()
print("n=%d rounds=%d active0=%d fwd device %.1f ms (%.1f MB/s) kernels %s" % (len(x), tm["rounds"], tm["active_after_round0"], tm["total_ms"], len(x) / 1e3 / tm["total_ms"], {k: round(v["ms"], 1) for k, v in tm["kernels"].items()}))
back = ctx.inverse(y)
ti = ctx.timings().as_dict()
print("roundtrip", bool(np.array_equal(back, x)), "inv device %.1f ms" % ti["total_ms"])
if len(x) <= (1 << 26):
    t0 = time.perf_counter(); want = O.forward(x); print("oracle %.1f s  equal=%s" % (time.perf_counter() - t0, bool(np.array_equal(want, y))))
