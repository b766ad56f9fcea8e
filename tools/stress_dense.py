"""Manual stress (GPU box): inputs of 4 .. 12 MiB built from repeated material -- many copies of short phrases (groups far larger
than the LDS cap), long runs, nested copies of long blocks, small alphabets -- so that the group-local rounds, their larger-group
path (two sorts, regrouping on both key words) and the quadrupled step all run; forward against the oracle, inverse back.
    python tools/stress_dense.py [cases] [seed0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for s in range(seed0, seed0 + cases):
    rng = np.random.default_rng(70000 + s)
    n = int(rng.integers(4 << 20, 12 << 20))
    sigma = int(rng.choice([2, 4, 20, 96, 256]))
    x = rng.integers(0, sigma, size=n, dtype=np.uint8)
    kind = s % 4
    if kind == 0:                                   # a few phrases pasted hundreds to thousands of times
        for _ in range(int(rng.integers(2, 8))):
            L = int(2 ** rng.uniform(3, 9)); ph = rng.integers(0, sigma, size=L, dtype=np.uint8)
            for at in rng.integers(0, n - L, size=int(rng.integers(300, 20000))): x[at:at + L] = ph
    elif kind == 1:                                 # long blocks copied a few times, nested, with point mutations
        for _ in range(int(rng.integers(3, 12))):
            L = int(2 ** rng.uniform(10, 21)); src = int(rng.integers(0, n - L)); dst = int(rng.integers(0, n - L))
            x[dst:dst + L] = x[src:src + L].copy()
        m = int(rng.integers(0, 200)); x[rng.integers(0, n, size=m)] = rng.integers(0, sigma, size=m, dtype=np.uint8)
    elif kind == 2:                                 # runs of one symbol of many lengths (groups of thousands that shrink slowly)
        for _ in range(int(rng.integers(200, 3000))):
            L = int(2 ** rng.uniform(2, 14)); at = int(rng.integers(0, n - L)); x[at:at + L] = rng.integers(0, sigma)
    else:                                           # periodic stretches with different periods
        for _ in range(int(rng.integers(20, 200))):
            per = rng.integers(0, sigma, size=int(rng.integers(1, 40)), dtype=np.uint8)
            L = int(2 ** rng.uniform(6, 18)); at = int(rng.integers(0, n - L)); x[at:at + L] = np.resize(per, L)
    y = ctx.forward(x)
    t = ctx.timings()
    want = O.forward(x)
    ok = np.array_equal(y, want) and np.array_equal(ctx.inverse(y), x)
    print("seed", s, "kind", kind, "n", n, "sigma", sigma, "rounds", t.rounds, "tied %.0f%%" % (100.0 * t.active_after_round0 / n), "OK" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
print("cases", cases, "bad", bad, "%.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
