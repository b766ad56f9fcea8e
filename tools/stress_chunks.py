"""Manual stress (GPU box) for the chunked rounds (csrc/chunk_rounds.h): inputs of 0.2 .. 3 MiB in which most positions stay tied after
round 0 (so the list is >= 65 536 elements and the chunk form runs), of the shapes that exercise its corners -- whole-input periods
(equal Lyndon factors: the rounds end on "no split"), hundreds of factors (the general-arithmetic instantiation), giant groups that
shrink into chunks over several rounds (big list -> appended chunks), nested copies, tiny alphabets; forward against the oracle,
inverse back.     python tools/stress_chunks.py [cases] [seed0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for s in range(seed0, seed0 + cases):
    rng = np.random.default_rng(90000 + s)
    n = int(2 ** rng.uniform(17.7, 21.5))
    sigma = int(rng.choice([2, 3, 4, 20, 96, 200]))
    kind = s % 6
    if kind == 0:                                   # u^k with a little noise or none: equal factors, equal rotations
        u = rng.integers(0, sigma, size=int(rng.integers(1000, 200000)), dtype=np.uint8)
        x = np.resize(u, n).copy()
        if s % 12 == 0:
            hits = rng.integers(0, n, size=int(rng.integers(1, 30))); x[hits] = rng.integers(0, sigma, size=hits.size, dtype=np.uint8)
    elif kind == 1:                                 # falling first letters: hundreds of Lyndon factors, each block there several times
        parts = []
        c = 250
        while sum(len(p) for p in parts) < n and c >= 0:
            body = rng.integers(c + 1, min(c + 1 + int(rng.integers(1, 5)), 256), size=int(rng.integers(200, 4000)), dtype=np.uint8)
            block = np.concatenate([np.array([c], dtype=np.uint8), body])
            parts += [block] * int(rng.integers(1, 4))
            c -= 1
        x = np.concatenate(parts)[:max(n, 1)]
    elif kind == 2:                                 # a few phrases pasted thousands of times into noise: groups of thousands
        x = rng.integers(0, sigma, size=n, dtype=np.uint8)
        for _ in range(int(rng.integers(2, 6))):
            L = int(2 ** rng.uniform(3, 8)); ph = rng.integers(0, sigma, size=L, dtype=np.uint8)
            for at in rng.integers(0, n - L, size=int(rng.integers(1000, 12000))): x[at:at + L] = ph
    elif kind == 3:                                 # nested copies of long blocks
        x = rng.integers(0, sigma, size=n, dtype=np.uint8)
        for _ in range(int(rng.integers(4, 14))):
            L = int(2 ** rng.uniform(9, np.log2(n) - 1.5)); src = int(rng.integers(0, n - L)); dst = int(rng.integers(0, n - L))
            x[dst:dst + L] = x[src:src + L].copy()
    elif kind == 4:                                 # runs and short periods over a tiny alphabet
        x = rng.integers(0, 2, size=n, dtype=np.uint8)
        for _ in range(int(rng.integers(50, 600))):
            per = rng.integers(0, 2, size=int(rng.integers(1, 9)), dtype=np.uint8)
            L = int(2 ** rng.uniform(4, 12)); at = int(rng.integers(0, n - L)); x[at:at + L] = np.resize(per, L)
    else:                                           # the text workload, several seeds and odd sizes
        x = O.generate("text", n, 100 + s)
    x = np.ascontiguousarray(x, dtype=np.uint8)
    t1 = time.time()
    want = O.forward(x)
    to = time.time() - t1
    y = ctx.forward(x)
    t = ctx.timings()
    ok = np.array_equal(y, want) and np.array_equal(ctx.inverse(y), x)
    print("seed", s, "kind", kind, "n", len(x), "sigma", sigma, "factors", t.factors, "rounds", t.rounds, "tied %.0f%%" % (100.0 * t.active_after_round0 / len(x)),
          "oracle %.1f s" % to, "OK" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
    if time.time() - t0 > float(os.environ.get("STRESS_BUDGET_S", "400")): print("time budget reached after", s - seed0 + 1, "cases"); break
print("bad", bad, "%.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
