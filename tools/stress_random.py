"""Manual stress (GPU box): many random structured inputs of widely varying sizes, forward/inverse against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for s in range(seed0, seed0 + cases):
    rng = np.random.default_rng(50000 + s)
    n = int(2 ** rng.uniform(0, 22.5))
    sigma = int(rng.choice([1, 2, 3, 4, 5, 16, 100, 256]))
    kind = s % 5
    if kind == 0:
        p = rng.random(sigma) ** rng.uniform(0.5, 6) + 1e-4
        x = rng.choice(sigma, size=n, p=p / p.sum()).astype(np.uint8)
    elif kind == 1:
        per = rng.integers(0, sigma, size=int(rng.integers(1, max(2, min(n, 500)))), dtype=np.uint8)
        x = np.resize(per, n).copy()
        k = int(n * rng.uniform(0, 0.002))
        if k: x[rng.integers(0, n, size=k)] = rng.integers(0, sigma, size=k, dtype=np.uint8)
    elif kind == 2:
        blk = rng.integers(0, sigma, size=max(1, n // int(rng.integers(2, 9))), dtype=np.uint8)
        x = np.resize(blk, n).copy()
        x[: min(n, 37)] = rng.integers(0, sigma, size=min(n, 37), dtype=np.uint8)
    elif kind == 3:
        n = min(n, 1 << 17)          # the oracle (= the reference's sequential fix-up loop) is quadratic on long runs
        x = np.sort(rng.integers(0, sigma, size=n, dtype=np.uint8))
        if rng.random() < 0.5: x = x[::-1].copy()
        c = int(rng.integers(0, n)); x = np.concatenate([x[c:], x[:c]])
    else:
        x = O.generate(["zipf", "dna", "uniform256"][s % 3], n, s)
    x = (x.astype(np.uint16) + int(rng.integers(0, 256 - min(sigma, 255)))).astype(np.uint8) if kind != 4 else x
    t1 = time.time(); y = ctx.forward(x); tf = time.time() - t1
    t1 = time.time(); want = O.forward(x); to = time.time() - t1
    t1 = time.time(); back = ctx.inverse(y); ti = time.time() - t1
    if tf > 2 or to > 2 or ti > 2: print('slow seed', s, 'n', n, 'sigma', sigma, 'kind', kind, 'gpu fwd %.1f s  oracle fwd %.1f s  gpu inv %.1f s' % (tf, to, ti), 'rounds', ctx.timings().rounds, flush=True)
    ok = np.array_equal(y, want) and np.array_equal(back, x) and np.array_equal(ctx.inverse(x), O.inverse(x))
    if not ok:
        bad += 1
        print("MISMATCH seed", s, "n", n, "sigma", sigma, "kind", kind, flush=True)
    if (s - seed0) % 25 == 24: print("done", s - seed0 + 1, "cases, bad", bad, "%.0f s" % (time.time() - t0), flush=True)
print("cases", cases, "bad", bad)
sys.exit(1 if bad else 0)
