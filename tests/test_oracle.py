"""CPU suite: the oracle against every golden vector the reference side provides.

Pins oracle/bwts_oracle.c: SURVEY.md 8(c) known answers (produced by the compiled reference),
outputs of the reference's own unbwts (tests/golden/ref_unbwts.json, made by make_golden.py),
the definition-level brute force, and -- where oracle/_ref/unbwts is present -- the live
reference inverse."""
import hashlib
import json
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(GOLD, "kat.json")))
REF = json.load(open(os.path.join(GOLD, "ref_unbwts.json")))


def fib_word(k):
    a, b = b"a", b"ab"
    for _ in range(k):
        a, b = b, b + a
    return b


ADVERSARIAL = [b"a", b"ab", b"ba", b"ba" * 100, b"ab" * 100, b"cba" * 70, b"a" * 257, bytes(range(256)),
               bytes(range(255, -1, -1)), fib_word(12), b"abcabcabd" * 30, b"\x00" * 10 + b"\xff" * 10, b"\xff\x00" * 40]


def test_generator_matches_survey():
    z = np.frombuffer(O.generate("uniform256", 24, 1).tobytes(), dtype="<u8")
    assert [hex(int(v)) for v in z] == KAT["generator_first_outputs_seed1"]
    assert O.generate("dna", 32, 1).tobytes() == b"CAATATCCGAAACGAGATGTCTGAGGAACACG"
    # seekable: a slice of the stream equals the stream's slice
    for kind in ("uniform256", "zipf", "dna", "text"):
        full = O.generate(kind, 1000, 5)
        assert np.array_equal(O.generate(kind, 300, 5, off=123), full[123:423])


def test_text_generator_pinned():
    """The repeat-rich text workload (oracle_generate kind 3): stream hash pinned, seekable deep into the stream, and the
    oracle's forward output on it accepted by the reference's own inverse (live, where oracle/_ref/unbwts exists)."""
    x = O.generate("text", 1 << 20, 1)
    assert hashlib.sha256(x.tobytes()).hexdigest() == "514bc95e4ca70b19bda5d6208945e12f666b52bcd760947f253559966bfc697d"
    assert np.array_equal(O.generate("text", 5000, 1, off=(1 << 20) - 5000), x[-5000:])
    y = O.forward(x)
    assert hashlib.sha256(y.tobytes()).hexdigest() == "58d139d1e7171e490c03786bff466c093ecae4e5a7754efe5f5613f99f019716"
    assert np.array_equal(O.inverse(y), x)
    if O.have_ref_unbwts():
        with tempfile.TemporaryDirectory() as td:
            assert np.array_equal(O.ref_unbwts(y, td), x)


def test_make_test_golden_is_the_oracles():
    """bijective-bwt_amd/testdata/testjunk.bwts (what `make test` compares the CLI's output with) is oracle_forward(testjunk)."""
    td = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bijective-bwt_amd", "testdata")
    x = np.fromfile(os.path.join(td, "testjunk"), dtype=np.uint8)
    y = np.fromfile(os.path.join(td, "testjunk.bwts"), dtype=np.uint8)
    assert x.size == y.size == 1 << 16
    assert np.array_equal(O.forward(x), y) and np.array_equal(O.inverse(y), x)


def test_known_answers():
    for text, bwts in KAT["text_to_bwts"]:
        assert O.forward(text.encode()).tobytes() == bwts.encode()
        assert O.forward_def(text.encode()).tobytes() == bwts.encode()
        assert O.inverse(bwts.encode()).tobytes() == text.encode()
    for text, bwts in KAT["hex_to_bwts"]:
        assert O.forward(bytes.fromhex(text)).tobytes().hex() == bwts
        assert O.inverse(bytes.fromhex(bwts)).tobytes().hex() == text
    asc = O.forward(bytes(range(256))).tobytes()
    assert hashlib.sha256(asc).hexdigest().startswith(KAT["bytes_asc_sha256_prefix"]) and asc[:4] == b"\xff\x00\x01\x02"
    desc = O.forward(bytes(range(255, -1, -1))).tobytes()
    assert hashlib.sha256(desc).hexdigest().startswith(KAT["bytes_desc_sha256_prefix"]) and desc[:3] == b"\x00\x01\x02"


@pytest.mark.parametrize("rec", [r for r in KAT["large"] if r["n"] <= 1 << 20], ids=lambda r: "%s-%d" % (r["kind"], r["n"]))
def test_large_hashes(rec):
    x = O.generate(rec["kind"], rec["n"], rec["seed"])
    assert hashlib.sha256(x.tobytes()).hexdigest() == rec["sha256_in"]
    y = O.forward(x)
    assert hashlib.sha256(y.tobytes()).hexdigest() == rec["sha256_bwts"]
    assert np.array_equal(O.inverse(y), x)


def test_large_hash_16MiB():
    rec = [r for r in KAT["large"] if r["n"] == 1 << 24 and r["kind"] == "zipf"][0]
    x = O.generate(rec["kind"], rec["n"], rec["seed"])
    assert hashlib.sha256(x.tobytes()).hexdigest() == rec["sha256_in"]
    assert hashlib.sha256(O.forward(x).tobytes()).hexdigest() == rec["sha256_bwts"]


def test_inverse_large_hash():
    rec = KAT["inverse_large"][0]
    x = O.inverse(O.generate(rec["kind"], rec["n"], rec["seed"]))
    assert hashlib.sha256(x.tobytes()).hexdigest() == rec["sha256_unbwts"]


def test_reference_unbwts_vectors_small():
    for rec in REF["small"]:
        y, x = bytes.fromhex(rec["bwts"]), bytes.fromhex(rec["text"])
        assert O.inverse(y).tobytes() == x          # inverse == reference program's output
        assert O.forward(x).tobytes() == y          # bijection: the same pair pins the forward bytes
        assert O.forward_def(x).tobytes() == y


@pytest.mark.parametrize("rec", REF["large"], ids=lambda r: "%s-%d" % (r["kind"], r["n"]))
def test_reference_unbwts_vectors_large(rec):
    y = O.generate(rec["kind"], rec["n"], rec["seed"])
    assert hashlib.sha256(y.tobytes()).hexdigest() == rec["sha256_bwts"]
    x = O.inverse(y)
    assert hashlib.sha256(x.tobytes()).hexdigest() == rec["sha256_text"]
    assert np.array_equal(O.forward(x), y)


def test_definition_vs_pipeline_random():
    rng = np.random.default_rng(11)
    for sigma in (1, 2, 3, 4, 256):
        for _ in range(120):
            n = int(rng.integers(1, 200))
            x = rng.integers(0, sigma, size=n, dtype=np.uint8)
            a, b = O.forward(x), O.forward_def(x)
            assert np.array_equal(a, b)
            assert np.array_equal(O.inverse(a), x)
            assert a[0] == x[-1]                     # mk_bwts_sa.c:188 invariant
            assert np.array_equal(O.forward(O.inverse(x)), x)   # any bytes are a valid inverse input


def test_64bit_index_instance_equals_32bit():
    """oracle_forward switches to 64-bit indices above the reference's range; the same code on small inputs must give the
    pinned 32-bit instance's bytes (this is what makes the golden above 2^31 in tests/golden/big_forward.json an oracle result)."""
    rng = np.random.default_rng(17)
    for x in ADVERSARIAL + [rng.integers(0, s, size=int(rng.integers(1, 3000)), dtype=np.uint8).tobytes() for s in (1, 2, 4, 256) for _ in range(10)]:
        assert np.array_equal(O.forward64(x), O.forward(x))
    for kind in ("zipf", "dna", "text"):
        x = O.generate(kind, 300001, 6)
        assert np.array_equal(O.forward64(x), O.forward(x))


def test_adversarial():
    for x in ADVERSARIAL:
        a, b = O.forward(x), O.forward_def(x)
        assert np.array_equal(a, b), x[:20]
        assert O.inverse(a).tobytes() == bytes(x)


def test_lyndon_starts_equal_prefix_minima_of_suffix_ranks():
    rng = np.random.default_rng(3)
    for x in ADVERSARIAL + [rng.integers(0, 3, size=500, dtype=np.uint8).tobytes() for _ in range(20)]:
        sa = O.suffix_array(x)
        isa = np.empty_like(sa)
        isa[sa] = np.arange(len(sa), dtype=sa.dtype)
        run = np.minimum.accumulate(isa)
        starts = [0] + [i for i in range(1, len(isa)) if isa[i] < run[i - 1]]
        assert list(O.lyndon_starts(x)) == starts


def test_suffix_array_bruteforce():
    rng = np.random.default_rng(5)
    for sigma in (1, 2, 4, 256):
        for _ in range(30):
            x = rng.integers(0, sigma, size=int(rng.integers(1, 120)), dtype=np.uint8).tobytes()
            want = sorted(range(len(x)), key=lambda i: x[i:])
            assert list(O.suffix_array(x)) == want


@pytest.mark.skipif(not O.have_ref_unbwts(), reason="oracle/_ref/unbwts not built (reference checkout absent)")
def test_live_reference_unbwts_pins_forward():
    """The reference's own inverse program must map oracle_forward(x) back to x."""
    rng = np.random.default_rng(9)
    with tempfile.TemporaryDirectory() as td:
        for x in [O.generate("zipf", 200001, 4), O.generate("dna", 65537, 4), rng.integers(0, 256, 5000, dtype=np.uint8),
                  np.frombuffer(b"ba" * 500, dtype=np.uint8), np.frombuffer(fib_word(15), dtype=np.uint8)]:
            y = O.forward(x)
            assert np.array_equal(O.ref_unbwts(y, td), x)
            assert np.array_equal(O.ref_unbwts(x, td), O.inverse(x))
