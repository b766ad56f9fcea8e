#!/usr/bin/env python3
"""Generates tests/golden/*.json.  Run in the build container only (needs oracle/_ref/unbwts,
i.e. the reference's own inverse program built from /root/reference/unbwts.c + map_file.c by
oracle/Makefile).  Nothing of the reference's source is stored: only inputs and outputs.

  kat.json          known-answer vectors recorded in SURVEY.md 8(c) (produced there by the
                    compiled reference); copied verbatim as data.
  ref_unbwts.json   outputs of the reference's unbwts on seeded inputs:
                    small cases as hex pairs, large cases as sha256 of generator streams.
                    Because the transform is a bijection and the reference's mk_bwts/unbwts
                    are mutual inverses, each pair (y -> x) is also a forward vector (x -> y).
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def main():
    if not O.have_ref_unbwts():
        raise SystemExit("oracle/_ref/unbwts missing: run `make -C oracle` where /root/reference exists")
    rng = np.random.default_rng(20261003)
    small = []
    with tempfile.TemporaryDirectory() as td:
        for sigma in (1, 2, 3, 4, 16, 256):
            for n in (1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233):
                y = rng.integers(0, sigma, size=n, dtype=np.uint8)
                if sigma <= 4:
                    y = np.frombuffer(b"acgt", dtype=np.uint8)[y]
                x = O.ref_unbwts(y, td)
                small.append({"bwts": y.tobytes().hex(), "text": x.tobytes().hex()})
        for name, y in (("ba*40", b"ba" * 40), ("ab*40", b"ab" * 40), ("a*64", b"a" * 64), ("desc", bytes(range(255, -1, -1))),
                        ("asc", bytes(range(256))), ("cba*30", b"cba" * 30)):
            x = O.ref_unbwts(y, td)
            small.append({"bwts": bytes(y).hex(), "text": x.tobytes().hex(), "name": name})
        large = []
        for kind, n, seed in (("uniform256", 1 << 20, 1), ("zipf", 1 << 20, 1), ("dna", 1 << 20, 1), ("zipf", 3000017, 5),
                              ("dna", 1 << 22, 2), ("uniform256", 1 << 22, 9)):
            y = O.generate(kind, n, seed)
            x = O.ref_unbwts(y, td)
            large.append({"kind": kind, "n": n, "seed": seed, "sha256_bwts": hashlib.sha256(y.tobytes()).hexdigest(),
                          "sha256_text": hashlib.sha256(x.tobytes()).hexdigest()})
    with open(os.path.join(HERE, "ref_unbwts.json"), "w") as f:
        json.dump({"producer": "oracle/_ref/unbwts (reference unbwts.c + map_file.c, unmodified, gcc -O2)",
                   "small": small, "large": large}, f, indent=0)
    print("wrote %d small + %d large vectors" % (len(small), len(large)))

    # the input / golden pair of `make -C bijective-bwt_amd test` (the reference's Makefile:30-33 compares against
    # testdata/testjunk.bwts, which it never shipped): 64 KiB of the text workload; the golden is the oracle's forward
    # output, accepted only if the reference's own unbwts turns it back into the input
    td_dir = os.path.join(os.path.dirname(os.path.dirname(HERE)), "bijective-bwt_amd", "testdata")
    os.makedirs(td_dir, exist_ok=True)
    x = O.generate("text", 1 << 16, 7)
    y = O.forward(x)
    with tempfile.TemporaryDirectory() as td:
        assert np.array_equal(O.ref_unbwts(y, td), x)
    x.tofile(os.path.join(td_dir, "testjunk"))
    y.tofile(os.path.join(td_dir, "testjunk.bwts"))
    print("wrote testdata/testjunk (+ .bwts)")


if __name__ == "__main__":
    main()
