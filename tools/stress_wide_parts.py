"""Manual stress (GPU box) for the n > 2^32 forward's rounds over the tied list (csrc/wide_path.h): the path forced onto inputs of
2 000 .. 400 000 bytes in which many positions stay tied after round 0, with random part sizes (64 .. 6 000 elements: dozens of parts per
round, later parts reading ranks the earlier ones refined) and tied-list blocks of 2^6 .. 2^12 pairs (survivors cross block borders);
forward against the oracle.  An input with a group larger than a part is refused (BWTS_E_RANGE) and counted, not failed.
    python tools/stress_wide_parts.py [cases] [seed0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BWTS_TEST_KNOBS"] = "1"; os.environ["BWTS_FORCE_WIDE"] = "2"; os.environ["BWTS_WIDE_SEG_LOG2"] = "13"; os.environ["BWTS_WIDE_DIRECT"] = "0"
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge
pkg = ge.load_package()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = refused = 0
t0 = time.time()
for s in range(seed0, seed0 + cases):
    rng = np.random.default_rng(70000 + s)
    n = int(2 ** rng.uniform(11, 18.6))
    sigma = int(rng.choice([2, 4, 20, 96, 200]))
    kind = s % 5
    if kind == 0:                                   # the text workload
        x = O.generate("text", n, 300 + s)
    elif kind == 1:                                 # nested copies of long blocks
        x = rng.integers(0, sigma, size=n, dtype=np.uint8)
        for _ in range(int(rng.integers(3, 12))):
            L = int(2 ** rng.uniform(6, np.log2(n) - 1.5)); src = int(rng.integers(0, n - L)); dst = int(rng.integers(0, n - L))
            x[dst:dst + L] = x[src:src + L].copy()
    elif kind == 2:                                 # a few phrases pasted many times into noise: groups of hundreds
        x = rng.integers(0, sigma, size=n, dtype=np.uint8)
        for _ in range(int(rng.integers(2, 6))):
            L = int(2 ** rng.uniform(3, 7)); ph = rng.integers(0, sigma, size=L, dtype=np.uint8)
            for at in rng.integers(0, n - L, size=int(rng.integers(20, 400))): x[at:at + L] = ph
    elif kind == 3:                                 # u^k with a little noise: whole-input period, equal rotations left at the end
        u = rng.integers(0, sigma, size=int(rng.integers(50, max(60, n // 3))), dtype=np.uint8)
        x = np.resize(u, n).copy()
        if s % 2: hits = rng.integers(0, n, size=int(rng.integers(1, 20))); x[hits] = rng.integers(0, sigma, size=hits.size, dtype=np.uint8)
    else:                                           # runs and short periods over a tiny alphabet
        x = rng.integers(0, 2, size=n, dtype=np.uint8)
        for _ in range(int(rng.integers(10, 200))):
            per = rng.integers(0, 2, size=int(rng.integers(1, 9)), dtype=np.uint8)
            L = int(2 ** rng.uniform(3, 9)); at = int(rng.integers(0, n - L)); x[at:at + L] = np.resize(per, L)
    x = np.ascontiguousarray(x, dtype=np.uint8)
    part = int(2 ** rng.uniform(6, 12.5)); lg = int(rng.integers(6, 13))
    while (64 << lg) < len(x): lg += 1              # (at most 64 blocks)
    os.environ["BWTS_WIDE_BUCKET"] = str(max(256, len(x) // int(rng.integers(2, 9))))
    os.environ["BWTS_WIDE_PART"] = str(part); os.environ["BWTS_WIDE_TBLOCK_LOG2"] = str(lg)
    want = O.forward(x)
    with pkg.Context(0) as ctx:
        try:
            y = ctx.forward(x)
        except Exception as e:
            code = getattr(e, "code", None)
            msg = str(e)
            if code == -5:                              # BWTS_E_RANGE
                refused += 1
                print("seed", s, "kind", kind, "n", len(x), "part", part, "refused:", msg, flush=True)
                continue
            raise
        t = ctx.timings()
        ok = np.array_equal(y, want)
    print("seed", s, "kind", kind, "n", len(x), "sigma", sigma, "part", part, "block", 1 << lg, "factors", t.factors, "rounds", t.rounds,
          "tied %.0f%%" % (100.0 * t.active_after_round0 / len(x)), "OK" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
    if time.time() - t0 > float(os.environ.get("STRESS_BUDGET_S", "400")): print("time budget reached after", s - seed0 + 1, "cases"); break
print("bad", bad, "refused", refused, "%.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
