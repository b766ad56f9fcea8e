"""Manual stress (GPU box): inverse transforms with a chosen number of elements in cycles that meet no splitter -- texts that open with a
descending run of m symbols (m one-symbol Lyndon factors, i.e. m LF cycles of their own) -- from a handful (named by arithmetic) over
hundreds and thousands (the per-class search) to most of the input (fallback to the index log); main and 64-bit paths.
    python tools/stress_unreached.py [log2n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log2n
bad = 0
for m in (0, 3, 17, 100, 300, 700, 1500, 4000, 20000, 100000, n // 4, n // 2):
    for seed in (1, 2):
        rng = np.random.default_rng(1000 * seed + m % 997)
        head = np.sort(rng.integers(0, 256, size=m, dtype=np.uint8))[::-1]            # non-increasing: every symbol a factor of its own (or equal neighbours: equal factors)
        tail = O.generate(["zipf", "uniform256"][seed % 2], n - m, seed + m)
        x = np.concatenate([head, tail]).astype(np.uint8)
        y = ctx.forward(x)
        back = ctx.inverse(y)
        t = ctx.timings()
        ok = np.array_equal(back, x)
        if n <= (1 << 22): ok = ok and np.array_equal(back, O.inverse(y))
        inv_raw = ctx.inverse(x)                                                      # and the inverse of the text itself (arbitrary bytes)
        ok = ok and np.array_equal(ctx.forward(inv_raw), x)
        print("m", m, "seed", seed, "cycles", t.factors, "unreached", t.unvisited, "OK" if ok else "MISMATCH", flush=True)
        bad += 0 if ok else 1
print("bad", bad)
sys.exit(1 if bad else 0)
