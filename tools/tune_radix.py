"""Manual tuning helper (GPU box): times the LSD passes for the tile shape selected by BWTS_RX_CONFIG."""
import os as _os; _os.environ.setdefault("BWTS_TEST_KNOBS", "1")      # this tool drives alternate-path knobs
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
m = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27
rng = np.random.default_rng(1)
k = rng.integers(0, 2**63, size=m, dtype=np.uint64)
v = np.arange(m, dtype=np.uint32)
# full stability check at a smaller size first
ks0 = rng.integers(0, 2**63, size=1 << 21, dtype=np.uint64) & np.uint64(0xFFFF00FF00FF)
a, b = ctx.debug_sort_pairs(ks0, np.arange(ks0.size, dtype=np.uint32), 64)
o = np.argsort(ks0, kind="stable")
assert np.array_equal(a, ks0[o]) and np.array_equal(b, o.astype(np.uint32)), "radix config is WRONG"
for rep in range(2):
    ks, vs = ctx.debug_sort_pairs(k, v, 64)
t = ctx.timings().as_dict()["kernels"]
ok = bool(np.all(ks[1:] >= ks[:-1]))
sc, hi = t["radix_scatter"], t["radix_hist"]
print("cfg=%s m=2^%d sorted=%s scatter %.3f ms/pass (%.0f GB/s alg)  hist %.3f ms/pass  scan %.3f" % (
    os.environ.get("BWTS_RX_CONFIG", "default"), int(np.log2(m)), ok, sc["ms"] / sc["launches"],
    sc["alg_bytes"] / 1e9 / (sc["ms"] / 1e3), hi["ms"] / hi["launches"], t["radix_scan"]["ms"] / t["radix_scan"]["launches"]))
