"""Manual check (GPU box): four lightly edited copies of the real-text corpus (like versions of the same files):
very long repeats, many doubling rounds.  Round trip only (the CPU oracle is too slow here)."""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
import __graft_entry__ as ge
import importlib.util
spec = importlib.util.spec_from_file_location("crt", "/root/repo/tools/check_realtext.py")
src = open("/root/repo/tools/check_realtext.py").read().split("limit = (1 <<")[0]
ns = {"__file__": "/root/repo/tools/check_realtext.py"}; exec(src, ns)
base = np.frombuffer(ns["corpus"](1 << 26), dtype=np.uint8)
rng = np.random.default_rng(7)
parts = []
for v in range(4):
    c = base.copy()
    k = len(c) // 5000
    c[rng.integers(0, len(c), size=k)] = rng.integers(32, 127, size=k, dtype=np.uint8)
    parts.append(c)
x = np.concatenate(parts)
pkg = ge.load_package(); ctx = pkg.Context(0); ctx.set_timing(2)
for rep in range(2):
    t0 = time.time(); y = ctx.forward(x); dt = time.time() - t0
tm = ctx.timings().as_dict()
print("versions x4: n=%d (%.0f MiB) rounds=%d active0=%.1f%% fwd device %.1f ms (%.0f MB/s) kernels %s" % (len(x), len(x)/2**20, tm["rounds"], 100.0*tm["active_after_round0"]/len(x), tm["total_ms"], len(x)/1e3/tm["total_ms"], {k: round(v["ms"],1) for k,v in tm["kernels"].items()}))
back = ctx.inverse(y); ti = ctx.timings().as_dict()
print("roundtrip", bool(np.array_equal(back, x)), "inv device %.1f ms (%.0f MB/s)" % (ti["total_ms"], len(x)/1e3/ti["total_ms"]))
