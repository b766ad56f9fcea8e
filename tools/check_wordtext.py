"""Manual check (GPU box): word-level Zipf text (repeated n-grams like natural language), forward/inverse timing and,
for small sizes, equality with the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import __graft_entry__ as ge

def wordtext(nbytes, seed, vocab=50000):
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    lens = rng.integers(2, 11, size=vocab)
    words = [letters[rng.choice(26, size=l, p=None)].tobytes() for l in lens]
    ranks = np.arange(1, vocab + 1, dtype=np.float64)
    p = (1.0 / ranks); p /= p.sum()
    out = bytearray()
    while len(out) < nbytes:
        idx = rng.choice(vocab, size=1 << 20, p=p)
        out += b" ".join(words[i] for i in idx) + b". "
    return np.frombuffer(bytes(out[:nbytes]), dtype=np.uint8)

pkg = ge.load_package(); ctx = pkg.Context(0); ctx.set_timing(2)
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
x = wordtext(1 << log2n, 3)
for rep in range(2):
    y = ctx.forward(x)
tm = ctx.timings().as_dict()
print("n=2^%d key_bits=%d rounds=%d active0=%d (%.1f%%) fwd device %.1f ms (%.0f MB/s) kernels %s" % (log2n, tm["key_bits"], tm["rounds"], tm["active_after_round0"], 100.0 * tm["active_after_round0"] / len(x), tm["total_ms"], len(x) / 1e3 / tm["total_ms"], {k: round(v["ms"], 1) for k, v in tm["kernels"].items()}))
back = ctx.inverse(y)
ti = ctx.timings().as_dict()
print("roundtrip", bool(np.array_equal(back, x)), "inv device %.1f ms (%.0f MB/s) cycles %d" % (ti["total_ms"], len(x) / 1e3 / ti["total_ms"], ti["factors"]))
if log2n <= 25:
    print("oracle equal:", bool(np.array_equal(O.forward(x), y)))
