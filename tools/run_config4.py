"""BASELINE config 4 on the GPU box: dna(n = 2^32, seed 1).  The reference cannot run this input (32-bit
indices), so parity is by properties: round trip = identity, byte histogram preserved, bwts[0] = T[n-1]."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kind = sys.argv[2] if len(sys.argv) > 2 else "dna"
n = 1 << log2n
ctx = pkg.Context(0)
d_in, d_out, d_back = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
ctx.generate(kind, 1, n, d_in)
res = {"workload": "%s(2^%d, seed 1)" % (kind, log2n)}
for rep in range(2):
    t0 = time.perf_counter(); ctx.forward_device(d_in, n, d_out); res["forward_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
tf = ctx.timings().as_dict()
res.update(factors=tf["factors"], rounds=tf["rounds"], key_symbols=tf["key_symbols"], key_bits=tf["key_bits"], active0=tf["active_after_round0"],
           forward_MBps=round(n / 1e6 / (res["forward_ms"] / 1e3), 1))
for rep in range(2):
    t0 = time.perf_counter(); ctx.inverse_device(d_out, n, d_back); res["inverse_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
ti = ctx.timings().as_dict()
res.update(cycles=ti["factors"], unvisited=ti["unvisited"], inverse_MBps=round(n / 1e6 / (res["inverse_ms"] / 1e3), 1))
res["roundtrip_exact"] = ctx.device_equal(d_in, d_back, n)
# histogram + first byte on a 64 MiB sample window plus ends (full download of 4 GiB x2 would be slow but fine: do it chunked)
hx = np.zeros(256, dtype=np.int64); hy = np.zeros(256, dtype=np.int64)
x_last = None; y_first = None
CH = 1 << 28
tmp = ctx.alloc(CH)
import ctypes
L = pkg.lib()
for off in range(0, n, CH):
    m = min(CH, n - off)
    a = np.empty(m, dtype=np.uint8); b = np.empty(m, dtype=np.uint8)
    L.bwts_copy_to_host(ctx._h, a.ctypes.data, d_in.ptr + off, m)
    L.bwts_copy_to_host(ctx._h, b.ctypes.data, d_out.ptr + off, m)
    hx += np.bincount(a, minlength=256); hy += np.bincount(b, minlength=256)
    if off == 0: y_first = int(b[0])
    x_last = int(a[-1])
res["histogram_preserved"] = bool(np.array_equal(hx, hy))
res["bwts0_is_last_text_byte"] = y_first == x_last
res["fwd_kernels_ms"] = {k: round(v["ms"], 2) for k, v in tf["kernels"].items()}
res["inv_kernels_ms"] = {k: round(v["ms"], 2) for k, v in ti["kernels"].items()}
print(json.dumps(res))
sys.exit(0 if res["roundtrip_exact"] and res["histogram_preserved"] and res["bwts0_is_last_text_byte"] else 1)
