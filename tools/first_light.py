"""Manual first-light script for the GPU box (not a pytest file)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
rng = np.random.default_rng(1)

def check_sort(m, bits):
    k = rng.integers(0, 2**63, size=m, dtype=np.uint64) >> np.uint64(64 - bits if bits < 64 else 0)
    v = np.arange(m, dtype=np.uint32)
    ks, vs = ctx.debug_sort_pairs(k, v, bits)
    order = np.argsort(k, kind="stable")
    ok = np.array_equal(ks, k[order]) and np.array_equal(vs, v[order])
    print("sort m=%d bits=%d ok=%s" % (m, bits, ok)); return ok

ok = True
for m, bits in [(1, 8), (5, 8), (1000, 16), (4096, 24), (4097, 64), (100000, 40), (1 << 20, 64)]:
    ok &= check_sort(m, bits)

def check_sa(x, name):
    sa = ctx.debug_suffix_array(x)
    want = O.suffix_array(x)
    good = np.array_equal(sa.astype(np.int64), want.astype(np.int64))
    print("sa %s n=%d ok=%s" % (name, len(x), good)); return good

def check_fwd(x, name):
    t = time.time(); y = ctx.forward(x); dt = time.time() - t
    want = O.forward(x) if len(x) > 2000 else O.forward_def(x)
    good = np.array_equal(y, want)
    tm = ctx.timings()
    back = ctx.inverse(y)
    rt = np.array_equal(back, np.frombuffer(bytes(x), dtype=np.uint8) if not isinstance(x, np.ndarray) else x)
    ti = ctx.timings()
    print("fwd %s n=%d ok=%s roundtrip=%s factors=%d rounds=%d lrounds=%d active0=%d fwd_ms=%.2f inv_ms=%.2f unvisited=%d" % (
        name, len(x), good, rt, tm.factors, tm.rounds, tm.lyndon_rounds, tm.active_after_round0, tm.total_ms, ti.total_ms, ti.unvisited))
    return good and rt

cases = [(b"banana", "banana"), (b"a", "a"), (b"ab" * 50, "(ab)^50"), (b"ba" * 50, "(ba)^50"), (b"a" * 100, "a^100"),
         (bytes(range(256)), "0..255"), (bytes(range(255, -1, -1)), "255..0"), (b"mississippi", "mississippi"),
         (b"cba" * 33, "(cba)^33")]
for x, name in cases:
    ok &= check_sa(x, name)
    ok &= check_fwd(x, name)
for kind in ("uniform256", "zipf", "dna"):
    for n in (1000, 5000, 70001, 1 << 20):
        x = O.generate(kind, n, 3)
        if n <= 70001: ok &= check_sa(x, kind)
        ok &= check_fwd(x, kind)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
