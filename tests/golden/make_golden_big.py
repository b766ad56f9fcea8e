#!/usr/bin/env python3
"""Generates tests/golden/big_forward.json: sha256 of the oracle's forward transform of inputs beyond the reference's
32-bit index range (mk_bwts_sa.c:26-27), computed with the oracle's 64-bit-index instance of the same pipeline
(oracle_forward switches to it from 2^31 - 1 bytes on; tests/test_oracle.py holds that instance against the pinned 32-bit
one on small inputs).  Run in the build container only: ~40 GiB of RAM and tens of minutes per entry.

    python tests/golden/make_golden_big.py [kind n seed] ...     (default: zipf 2^31+12345 seed 3)
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

# beyond the reference's index range, then BASELINE.json's configs 2 and 3 and the text workload at full size (32-bit instance)
CASES = [("zipf", (1 << 31) + 12345, 3), ("uniform256", 1 << 28, 1), ("zipf", 1 << 30, 1), ("text", 1 << 30, 1)]


def main():
    path = os.path.join(HERE, "big_forward.json")
    recs = json.load(open(path))["cases"] if os.path.exists(path) else []
    cases = CASES
    if len(sys.argv) > 1:
        a = sys.argv[1:]
        cases = [(a[i], int(a[i + 1]), int(a[i + 2])) for i in range(0, len(a), 3)]
    for kind, n, seed in cases:
        if any(r["kind"] == kind and r["n"] == n and r["seed"] == seed for r in recs):
            continue
        t0 = time.time()
        x = O.generate(kind, n, seed)
        print("generated", kind, n, "in %.0f s" % (time.time() - t0), flush=True)
        t0 = time.time()
        y = O.forward(x)
        print("forward in %.0f s" % (time.time() - t0), flush=True)
        assert y[0] == x[-1]
        recs.append({"kind": kind, "n": n, "seed": seed, "sha256_in": hashlib.sha256(x.tobytes()).hexdigest(),
                     "sha256_bwts": hashlib.sha256(y.tobytes()).hexdigest()})
        json.dump({"producer": "oracle/bwts_oracle.c forward (64-bit-index instance from 2^31 - 1 bytes on, the pinned 32-bit one below; tests/golden/make_golden_big.py)", "cases": recs},
                  open(path, "w"), indent=1)
        del x, y


if __name__ == "__main__":
    main()
