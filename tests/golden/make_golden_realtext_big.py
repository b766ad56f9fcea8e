#!/usr/bin/env python3
"""Generates tests/golden/realtext_1GiB.json: sha-256 of 2^30 bytes of the image's real text (tests/realtext.py corpus_big) and of the
oracle's forward transform of it (oracle/bwts_oracle.c, the pinned CPU restatement of mk_bwts_sa.c:114-195).  ~10-15 min, ~12 GB; run in the
build container (same image as the GPU boxes)."""
import hashlib, json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import oracle_lib as O
import realtext
t0 = time.time()
x = np.frombuffer(realtext.corpus_big(1 << 30), dtype=np.uint8)
print("corpus", x.size, "bytes in %.0f s" % (time.time() - t0), flush=True)
assert x.size == 1 << 30, "the image holds less than 1 GiB of such files"
sha_in = hashlib.sha256(x.tobytes()).hexdigest()
t0 = time.time()
y = O.forward(x)
print("oracle forward %.0f s" % (time.time() - t0), flush=True)
rec = {"producer": "tests/golden/make_golden_realtext_big.py (oracle/bwts_oracle.c forward)", "n": int(x.size), "sigma": int(len(np.unique(x))),
       "sha256_in": sha_in, "sha256_bwts": hashlib.sha256(y.tobytes()).hexdigest(),
       "sha256_first_2p26": hashlib.sha256(x[: 56242662].tobytes()).hexdigest()}
json.dump(rec, open(os.path.join(HERE, "realtext_1GiB.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
