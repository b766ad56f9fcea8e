#!/usr/bin/env python3
"""Generates tests/golden/realtext.json: sha-256 of the image's real-text corpus (tests/realtext.py, limit 2^26) and of the oracle's
forward transform of it (oracle/bwts_oracle.c, the pinned CPU restatement of mk_bwts_sa.c:114-195).  ~5 s; run in the build container."""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import oracle_lib as O
import realtext
x = np.frombuffer(realtext.corpus(1 << 26), dtype=np.uint8)
y = O.forward(x)
assert np.array_equal(O.inverse(y), x)
json.dump({"producer": "tests/golden/make_golden_realtext.py (oracle/bwts_oracle.c forward)", "limit_log2": 26, "n": int(x.size),
           "sigma": int(len(np.unique(x))), "sha256_in": hashlib.sha256(x.tobytes()).hexdigest(),
           "sha256_bwts": hashlib.sha256(y.tobytes()).hexdigest()}, open(os.path.join(HERE, "realtext.json"), "w"), indent=1)
print(open(os.path.join(HERE, "realtext.json")).read())
