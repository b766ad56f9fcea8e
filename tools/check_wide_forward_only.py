"""Manual check (GPU box): the forward transform of an input too large for its inverse to fit beside it (dna of 16 .. 48 GiB): the n > 2^32
path without the rank array (few ties: wide_direct_groups_kernel).  No oracle and no device inverse exist at this size, so the output is
checked on the host through what the transform must satisfy: it permutes the input's bytes; bwts[0] = T[n-1] (mk_bwts_sa.c:188); and an
LF walk (unbwts.c:50-52, 66-82, exact ranks from block counts) of `steps` steps from a slot in the middle of the output reads some
stretch of the text backwards -- the walked bytes, reversed, are searched for in the input and must be there.
    python tools/check_wide_forward_only.py [gib] [kind] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 24
kind = sys.argv[2] if len(sys.argv) > 2 else "dna"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
n = gib << 30
ctx = pkg.Context(0); ctx.set_timing(2)
a, b = ctx.alloc(n), ctx.alloc(n)
ctx.generate(kind, 1, n, a)
t0 = time.perf_counter(); ctx.forward_device(a, n, b); dt = time.perf_counter() - t0
k = ctx.timings().as_dict()
print("forward %.0f ms wall (first call), device %.0f ms" % (1e3 * dt, k["total_ms"]), {x: round(v["ms"], 1) for x, v in k["kernels"].items()},
      "rounds", k["rounds"], "tied", k["active_after_round0"], "factors", k["factors"], "device GiB %.0f" % (k["device_bytes"] / 2**30), flush=True)
t0 = time.perf_counter(); ctx.forward_device(a, n, b); dt = time.perf_counter() - t0
print("forward again (arenas in place): %.0f ms wall = %.2f GB/s" % (1e3 * dt, n / 1e9 / dt), flush=True)
x, y = a.download(), b.download()
a.free(); b.free()
hx, hy = np.zeros(256, np.int64), np.zeros(256, np.int64)
B = 1 << 26
for o in range(0, n, B):
    hx += np.bincount(x[o:o + B], minlength=256); hy += np.bincount(y[o:o + B], minlength=256)
print("histograms equal:", bool(np.array_equal(hx, hy)), " bwts[0] == T[n-1]:", bool(y[0] == x[-1]), flush=True)
# LF walk with exact ranks: per symbol, counts per block of 2^20
syms = np.flatnonzero(hy)
Bk = 1 << 20
nb = (n + Bk - 1) // Bk
cum = {}
for c in syms:
    per = np.zeros(nb + 1, dtype=np.int64)
    for o in range(0, n, B):
        blk = (y[o:o + B] == c)
        cnt = np.add.reduceat(blk.view(np.uint8), np.arange(0, blk.size, Bk), dtype=np.int64)
        per[o // Bk + 1: o // Bk + 1 + cnt.size] = cnt
    cum[int(c)] = np.cumsum(per)
C = np.concatenate([[0], np.cumsum(hy)])[:256]
r = n // 2 + 12345
got = bytearray()
for t in range(steps):
    c = int(y[r]); got.append(c)
    bk = r // Bk
    r = int(C[c] + cum[c][bk] + np.count_nonzero(y[bk * Bk:r] == c))
pat = bytes(got[::-1])
t0 = time.perf_counter()
found = -1
xb = memoryview(x)
CH = 1 << 30
for o in range(0, n, CH):                                    # search chunk by chunk (with an overlap of the pattern's length)
    hay = xb[max(0, o - len(pat)): o + CH].tobytes()
    i = hay.find(pat)
    if i >= 0: found = max(0, o - len(pat)) + i; break
print("LF walk of %d steps from slot %d: the bytes read are the text backwards from position %d (%.0f s search)" % (steps, n // 2 + 12345, found + len(pat), time.perf_counter() - t0)
      if found >= 0 else "LF walk: the bytes read are NOT in the text", flush=True)
ok = bool(np.array_equal(hx, hy)) and bool(y[0] == x[-1]) and found >= 0
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
