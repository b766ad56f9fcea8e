"""GPU suite (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and -- at BASELINE.json's sizes -- through
size-independent properties.  Integer/byte work: every comparison is bit-exact."""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_lib as O
from test_oracle import ADVERSARIAL, KAT, REF, fib_word

BIG = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_forward.json")))

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bijective-bwt_amd")


# ---------------------------------------------------------------------------------------------
# kernel-level
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,bits", [(1, 8), (2, 1), (63, 8), (64, 16), (65, 64), (4095, 24), (4096, 32), (4097, 64),
                                    (100003, 40), (1 << 20, 64), (3 << 20, 63), (10_000_000, 48)])
def test_radix_sort_pairs_stable(ctx, m, bits):
    rng = np.random.default_rng(m * 131 + bits)
    k = rng.integers(0, 2**63, size=m, dtype=np.uint64)
    if bits < 64:
        k &= np.uint64((1 << bits) - 1)
    if m > 1000:   # heavy duplicates exercise stability and the skewed-digit paths
        k[: m // 2] = k[0] & np.uint64(0xFF00FF)
    v = np.arange(m, dtype=np.uint32)
    ks, vs = ctx.debug_sort_pairs(k, v, bits)
    order = np.argsort(k, kind="stable")
    assert np.array_equal(ks, k[order])
    assert np.array_equal(vs, v[order])


@pytest.mark.parametrize("m", [100_000, 1_000_000])
def test_radix_sort_pairs_runs_of_equal_digits(ctx, m):
    """Keys that are already sorted, and keys from a handful of values: every pass sees long runs of equal digits in neighbouring
    lanes -- what the histogram sweeps add up run by run (hist_add_runs) instead of lane by lane."""
    rng = np.random.default_rng(m)
    for k in (np.sort(rng.integers(0, 2**40, size=m, dtype=np.uint64)),
              rng.choice(np.array([3, 3 << 8, 0x0102030405, 7 << 32], dtype=np.uint64), size=m),
              np.repeat(rng.integers(0, 2**40, size=(m + 99) // 100, dtype=np.uint64), 100)[:m]):
        v = np.arange(m, dtype=np.uint32)
        ks, vs = ctx.debug_sort_pairs(k, v, 40)
        order = np.argsort(k, kind="stable")
        assert np.array_equal(ks, k[order]) and np.array_equal(vs, v[order])


def _inputs_small():
    rng = np.random.default_rng(77)
    out = [(name, np.frombuffer(bytes(x), dtype=np.uint8)) for name, x in
           [("adv%d" % i, a) for i, a in enumerate(ADVERSARIAL)]]
    for sigma in (1, 2, 3, 4, 256):
        for n in (1, 2, 7, 64, 65, 199, 2049, 4097):
            out.append(("rand-s%d-n%d" % (sigma, n), rng.integers(0, sigma, size=n, dtype=np.uint8)))
    for kind in ("uniform256", "zipf", "dna"):
        for n in (1000, 70001):
            out.append(("%s-%d" % (kind, n), O.generate(kind, n, 3)))
    out.append(("fib20", np.frombuffer(fib_word(20), dtype=np.uint8)))
    out.append(("period-long", np.frombuffer((b"abcab" * 5000) + b"b", dtype=np.uint8)))
    out.append(("runs", np.frombuffer(b"".join(bytes([c]) * 3000 for c in (5, 4, 9, 4, 5, 1)), dtype=np.uint8)))
    return out


SMALL = _inputs_small()


@pytest.mark.parametrize("name,x", SMALL, ids=[s[0] for s in SMALL])
def test_suffix_array_and_lyndon_vs_oracle(ctx, name, x):
    assert np.array_equal(ctx.debug_suffix_array(x).astype(np.int64), O.suffix_array(x).astype(np.int64))
    assert np.array_equal(ctx.debug_lyndon(x).astype(np.int64), O.lyndon_starts(x))


# ---------------------------------------------------------------------------------------------
# transform-level parity
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,x", SMALL, ids=[s[0] for s in SMALL])
def test_forward_inverse_vs_oracle_small(ctx, name, x):
    y = ctx.forward(x)
    assert np.array_equal(y, O.forward(x))
    if x.size <= 300:
        assert np.array_equal(y, O.forward_def(x))
    assert y[0] == x[-1]                                    # mk_bwts_sa.c:188
    assert np.array_equal(ctx.inverse(y), x)                # unbwts o mk_bwts = id
    assert np.array_equal(ctx.inverse(x), O.inverse(x))     # any bytes are a valid inverse input
    assert np.array_equal(ctx.forward(ctx.inverse(x)), x)   # mk_bwts o unbwts = id


def test_known_answers_through_cabi(ctx):
    for text, bwts in KAT["text_to_bwts"]:
        assert ctx.forward(text.encode()).tobytes() == bwts.encode()
        assert ctx.inverse(bwts.encode()).tobytes() == text.encode()
    for text, bwts in KAT["hex_to_bwts"]:
        assert ctx.forward(bytes.fromhex(text)).tobytes().hex() == bwts
        assert ctx.inverse(bytes.fromhex(bwts)).tobytes().hex() == text
    assert hashlib.sha256(ctx.forward(bytes(range(256))).tobytes()).hexdigest().startswith(KAT["bytes_asc_sha256_prefix"])
    assert hashlib.sha256(ctx.forward(bytes(range(255, -1, -1))).tobytes()).hexdigest().startswith(KAT["bytes_desc_sha256_prefix"])


def test_reference_unbwts_vectors_through_cabi(ctx):
    for rec in REF["small"]:
        y, x = bytes.fromhex(rec["bwts"]), bytes.fromhex(rec["text"])
        assert ctx.inverse(y).tobytes() == x
        assert ctx.forward(x).tobytes() == y


@pytest.mark.parametrize("rec", KAT["large"], ids=lambda r: "%s-%d" % (r["kind"], r["n"]))
def test_large_golden_hashes(ctx, rec):
    """Hashes produced by the compiled reference, inputs generated on the device."""
    n = rec["n"]
    d_in, d_out, d_back = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    try:
        ctx.generate(rec["kind"], rec["seed"], n, d_in)
        assert hashlib.sha256(d_in.download().tobytes()).hexdigest() == rec["sha256_in"]   # device generator == spec
        ctx.forward_device(d_in, n, d_out)
        assert hashlib.sha256(d_out.download().tobytes()).hexdigest() == rec["sha256_bwts"]
        ctx.inverse_device(d_out, n, d_back)
        assert ctx.device_equal(d_in, d_back, n)
    finally:
        for b in (d_in, d_out, d_back):
            b.free()


@pytest.mark.parametrize("rec", REF["large"], ids=lambda r: "%s-%d" % (r["kind"], r["n"]))
def test_large_reference_unbwts_hashes(ctx, rec):
    y = O.generate(rec["kind"], rec["n"], rec["seed"])
    x = ctx.inverse(y)
    assert hashlib.sha256(x.tobytes()).hexdigest() == rec["sha256_text"]
    assert np.array_equal(ctx.forward(x), y)


def test_inverse_attempts_reported(ctx):
    """bwts_timings.attempts: the cycle walk runs once on natural inputs; a constant input (every element its own LF cycle, nearly all
    of them unreached by any splitter walk) needs the index log: two attempts."""
    x = O.generate("zipf", 3 << 20, 5)
    assert np.array_equal(ctx.inverse(x), O.inverse(x)) and ctx.timings().attempts == 1
    z = np.full(6 << 20, 7, dtype=np.uint8)
    assert np.array_equal(ctx.inverse(z), z) and 1 <= ctx.timings().attempts <= 5


@pytest.mark.parametrize("kind,n,seed", [("zipf", 4 << 20, 11), ("dna", 3000017, 12), ("uniform256", (2 << 20) + 5, 13)])
def test_mid_size_vs_oracle(ctx, kind, n, seed):
    x = O.generate(kind, n, seed)
    y = ctx.forward(x)
    assert np.array_equal(y, O.forward(x))
    assert np.array_equal(ctx.inverse(y), x)


@pytest.mark.parametrize("kind,n", [("zipf", 65535), ("zipf", 65536), ("dna", 65537), ("uniform256", 73001), ("zipf", (1 << 22) - 1), ("dna", (1 << 22) + 1)])
def test_threshold_sizes_vs_oracle(ctx, kind, n):
    """Sizes on either side of the engine's path switches (packed round-0 sort from 65536, binned rank build from 2^22)."""
    x = O.generate(kind, n, 21)
    y = ctx.forward(x)
    assert np.array_equal(y, O.forward(x))
    assert np.array_equal(ctx.inverse(y), x)


def test_dense_ties_at_bin_threshold_vs_oracle(ctx):
    """Exactly 2^22 bytes, nearly everything tied after round 0: first size the binned dense-rank build takes."""
    block = O.generate("zipf", 1 << 20, 8)
    x = np.concatenate([block, block, block, block])
    assert len(x) == 1 << 22
    y = ctx.forward(x)
    assert ctx.timings().active_after_round0 > len(x) // 32
    assert np.array_equal(y, O.forward(x))
    assert np.array_equal(ctx.inverse(y), x)


def _structured_input(seed):
    """Random inputs with structure: skewed alphabets of several sizes, noisy periodic text, concatenated repeats."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(66000, 400000))
    kind = seed % 4
    sigma = int(rng.choice([2, 3, 4, 16, 64, 256]))
    if kind == 0:                                   # skewed i.i.d. symbols
        p = rng.random(sigma) ** 3 + 1e-3
        x = rng.choice(sigma, size=n, p=p / p.sum()).astype(np.uint8)
    elif kind == 1:                                 # a short period with sparse noise: deep ties, many equal rotations
        period = rng.integers(0, sigma, size=int(rng.integers(3, 200)), dtype=np.uint8)
        x = np.resize(period, n).copy()
        hits = rng.integers(0, n, size=n // 2000)
        x[hits] = rng.integers(0, sigma, size=hits.size, dtype=np.uint8)
    elif kind == 2:                                 # long repeats of a random block, different offsets
        block = rng.integers(0, sigma, size=n // 5, dtype=np.uint8)
        parts = [block, block[: len(block) // 2], rng.integers(0, sigma, size=777, dtype=np.uint8), block[len(block) // 3:], block]
        x = np.concatenate(parts)
    else:                                           # descending runs: many Lyndon factors
        x = np.sort(rng.integers(0, sigma, size=n, dtype=np.uint8))[::-1].copy()
        cut = int(rng.integers(1, n))
        x = np.concatenate([x[cut:], x[:cut]])
    return x + np.uint8(rng.integers(0, 256 - sigma + 1))     # anywhere in the byte range


@pytest.mark.parametrize("seed", range(16))
def test_structured_random_inputs_vs_oracle(ctx, seed):
    x = _structured_input(seed)
    y = ctx.forward(x)
    assert np.array_equal(y, O.forward(x))
    assert np.array_equal(ctx.inverse(y), x)
    assert np.array_equal(ctx.forward(ctx.inverse(x)), x)       # the inverse of arbitrary bytes, and back


def test_deep_repeats_vs_oracle(ctx):
    """Long repeats force many doubling rounds and a large active set (real-text shape)."""
    rng = np.random.default_rng(5)
    block = rng.integers(97, 101, size=50000, dtype=np.uint8)
    x = np.concatenate([block, block[:40000], rng.integers(97, 101, size=1000, dtype=np.uint8), block[10000:], block])
    y = ctx.forward(x)
    # (repeats of 50 000 symbols.  The classic rounds need log4(50 000 / 4) + 1 >= 6 of them; with parked chains -- csrc/chunk_rounds.h --
    # the copies settle as soon as the END of a repeat does, so fewer are enough; several are still needed)
    assert ctx.timings().rounds >= 3
    assert np.array_equal(y, O.forward(x))
    assert np.array_equal(ctx.inverse(y), x)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_dense_rounds_repeated_material(ctx, kind):
    """4 MiB built from repeated material (tools/stress_dense.py): phrases pasted thousands of times (groups far larger than the LDS
    cap: two sorts and regrouping on both key words), nested block copies, runs of one symbol, periodic stretches."""
    rng = np.random.default_rng(71000 + kind)
    n, sigma = 4 << 20, [4, 96, 2, 20][kind]
    x = rng.integers(0, sigma, size=n, dtype=np.uint8)
    if kind == 0:
        for _ in range(5):
            L = int(2 ** rng.uniform(3, 9)); ph = rng.integers(0, sigma, size=L, dtype=np.uint8)
            for at in rng.integers(0, n - L, size=int(rng.integers(300, 20000))): x[at:at + L] = ph
    elif kind == 1:
        for _ in range(8):
            L = int(2 ** rng.uniform(10, 20)); src = int(rng.integers(0, n - L)); dst = int(rng.integers(0, n - L))
            x[dst:dst + L] = x[src:src + L].copy()
        x[rng.integers(0, n, size=100)] = rng.integers(0, sigma, size=100, dtype=np.uint8)
    elif kind == 2:
        for _ in range(1500):
            L = int(2 ** rng.uniform(2, 14)); at = int(rng.integers(0, n - L)); x[at:at + L] = rng.integers(0, sigma)
    else:
        for _ in range(100):
            per = rng.integers(0, sigma, size=int(rng.integers(1, 40)), dtype=np.uint8)
            L = int(2 ** rng.uniform(6, 18)); at = int(rng.integers(0, n - L)); x[at:at + L] = np.resize(per, L)
    y = ctx.forward(x)
    assert ctx.timings().active_after_round0 > n // 64
    assert np.array_equal(y, O.forward(x))
    assert np.array_equal(ctx.inverse(y), x)


def test_chunk_rounds_many_factors_equal_rotations(ctx):
    """More Lyndon factors than the chunk kernel keeps in LDS (its general-arithmetic instantiation), every factor there twice: all
    positions stay tied to the end (equal rotations of equal factors), so the rounds stop on "no group split" and the leftover groups are
    laid out by chunk_rest_kernel."""
    rng = np.random.default_rng(4242)
    parts = []
    for c in range(200, -1, -1):                    # a word whose first letter is strictly its smallest is a Lyndon word; first letters fall
        body = rng.integers(c + 1, min(c + 6, 256), size=int(rng.integers(2500, 3500)), dtype=np.uint8)
        block = np.concatenate([np.array([c], dtype=np.uint8), body])
        parts += [block, block]
    x = np.concatenate(parts)
    y = ctx.forward(x)
    t = ctx.timings()
    assert t.factors == 402 and t.active_after_round0 == len(x)
    assert np.array_equal(y, O.forward(x))
    assert np.array_equal(ctx.inverse(y), x)


def test_chunk_rounds_groups_of_hundreds_and_thousands(ctx):
    """Phrases pasted 300 ... 6000 times into noise: after round 0 the tied list has groups of a few hundred members (WIDE chunks: a
    tile ordered by the workgroup's bitonic sort), of up to 2048 (a tile of their own) and of several thousand (the big list, whose
    pieces leave for chunks as they shrink) next to pairs and triples; several sizes of the whole, two alphabets."""
    for seed, n, sigma in ((77, 1 << 21, 96), (78, (1 << 20) + 12345, 4), (79, 3 << 20, 200)):
        rng = np.random.default_rng(seed)
        x = rng.integers(0, sigma, size=n, dtype=np.uint8)
        for times in (300, 700, 1500, 2048, 2049, 3000, 6000):
            L = int(rng.integers(24, 90))
            ph = rng.integers(0, sigma, size=L, dtype=np.uint8)
            for at in rng.integers(0, n - L, size=times):
                x[at:at + L] = ph
        y = ctx.forward(x)
        t = ctx.timings()
        assert t.active_after_round0 >= 1 << 16                   # (the chunk form takes lists from 65 536 elements on)
        assert np.array_equal(y, O.forward(x)), (seed, n, sigma)
        assert np.array_equal(ctx.inverse(y), x)


def test_dense_ties_large_vs_oracle(ctx):
    """n >= 2^22 with most elements tied after round 0: the dense rank array is built by the binned scatter."""
    block = O.generate("zipf", 1 << 21, 5)
    x = np.concatenate([block, block, O.generate("zipf", 1000, 6), block[: 1 << 20], block])      # ~7.3 MiB, long repeats
    y = ctx.forward(x)
    t = ctx.timings()
    assert t.active_after_round0 > len(x) // 32 and t.rounds >= 3      # (classic rounds: >= 6; parked chains settle a repeat when its end settles)
    assert np.array_equal(y, O.forward(x))
    assert np.array_equal(ctx.inverse(y), x)


def test_inverse_many_tiny_cycles(ctx):
    """Theta(n) LF cycles without a splitter: found and resolved on the device, sized by their number."""
    x = np.sort(O.generate("zipf", 3 << 20, 2))          # sorted bytes: LF is the identity, n cycles of length 1
    got = ctx.inverse(x)
    assert ctx.timings().unvisited > x.size // 2         # only the splitters themselves were reached by a walk
    assert np.array_equal(got, O.inverse(x))
    y = np.frombuffer(b"ba" * (1 << 20), dtype=np.uint8)  # forward gives 2^20 factors of length 2
    f = ctx.forward(y)
    assert np.array_equal(f, O.forward(y))
    assert np.array_equal(ctx.inverse(f), y)
    z = np.frombuffer(b"abc" * 700001, dtype=np.uint8)    # as inverse input: LF cycles of assorted small lengths
    assert np.array_equal(ctx.inverse(z), O.inverse(z))
    assert np.array_equal(ctx.forward(ctx.inverse(z)), z)


def test_no_room_for_the_big_list_falls_back_to_the_tile_form(pkg, capfd):
    """Groups of thousands of members go to the chunk rounds' big list, whose buffers are reserved once the list's size is known -- after
    the first two blocks.  Without room for them the tile form takes over, as it does when the first blocks do not fit (nothing but
    the function's own buffers has been written by then): same bytes, no error.  BWTS_BIGLIST_NOMEM=1 refuses that reservation."""
    rng = np.random.default_rng(5)
    phrase = rng.integers(97, 123, 200, dtype=np.uint8)
    x = np.concatenate([np.tile(phrase, 5000), O.generate("zipf", 1 << 20, 4)])
    want = O.forward(x)
    saved = {k: os.environ.get(k) for k in ("BWTS_TEST_KNOBS", "BWTS_BIGLIST_NOMEM", "BWTS_ROUND_TRACE")}
    try:
        os.environ.update(BWTS_TEST_KNOBS="1", BWTS_BIGLIST_NOMEM="1", BWTS_ROUND_TRACE="1")
        with pkg.Context(0) as ctx:                      # (a context reads its switches when it is made)
            got = ctx.forward(x)
            assert "no room for the big list" in capfd.readouterr().err          # (the input does have one, and it was refused)
            assert np.array_equal(got, want)
            assert np.array_equal(ctx.inverse(got), x)
    finally:
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    with pkg.Context(0) as ctx:
        assert np.array_equal(ctx.forward(x), want)


def _short_factors_text(words, length, tail, seed):
    """`words` distinct Lyndon words of `length` bytes in decreasing order (each starts with its only smallest byte), then one long
    Lyndon factor (a single 0, then bytes 1..255): the transform has `words` LF cycles of `length` elements whose ranks lie scattered
    among the long factor's rotations -- the short leading factors of real text, in numbers."""
    rng = np.random.default_rng(seed)
    ws = set()
    while len(ws) < words:
        c = int(rng.integers(60, 200))
        ws.add(bytes([c]) + rng.integers(c + 1, 256, length - 1, dtype=np.uint8).tobytes())
    body = b"".join(sorted(ws, reverse=True))
    return np.frombuffer(body + b"\0" + rng.integers(1, 256, tail, dtype=np.uint8).tobytes(), dtype=np.uint8)


def test_inverse_unreached_named_by_their_cycles(ctx):
    """Several hundred unreached elements in short cycles (at this length every 16th element is a splitter: a cycle of 20 meets none
    with probability 0.27): dozens of residue classes miss three or more elements, which the classes' moments cannot name -- the
    cycles of the elements they do name can (moments_resolve_kernel).  One walk, not two (bwts_timings.attempts): the 1 GiB of real
    text used to take two.  With many more such elements few classes name one: the chase or the index log takes over, exact all the same."""
    x = _short_factors_text(150, 20, 1 << 20, 7)
    y = O.forward(x)
    got = ctx.inverse(y)
    t = ctx.timings()
    assert np.array_equal(got, x)
    assert t.unvisited >= 300 and t.factors == 151 and t.attempts == 1
    x = _short_factors_text(40, 9, (1 << 21) + 77, 8)              # a few hundred: the arithmetic alone, or one round of cycles
    assert np.array_equal(ctx.inverse(O.forward(x)), x) and ctx.timings().attempts == 1
    x = _short_factors_text(600, 24, 1 << 20, 9)                   # several per class
    got = ctx.inverse(O.forward(x))
    assert np.array_equal(got, x) and ctx.timings().factors == 601


def test_inverse_low_entropy_large(pkg):
    """Constant and sorted inputs of 96 / 48 MiB: LF is (close to) the identity, nearly every element sits in a cycle
    without a splitter.  The reference handles any bytes in 4 n memory (unbwts.c:45-52); the engine must stay within a
    bounded multiple of n on the device and use no host memory proportional to n."""
    with pkg.Context(0) as ctx:                          # a context of its own: device_bytes then counts this call's memory only
        n = 96 << 20
        x = np.full(n, 65, dtype=np.uint8)
        got = ctx.inverse(x)
        t = ctx.timings()
        assert np.array_equal(got, x)                    # one symbol: every cycle has length 1, the text is the input
        assert t.unvisited > n // 2 and t.factors == n
        assert t.device_bytes < 80 * n                   # (the earlier fallback -- every element a splitter -- needed 129 n)
        n = 48 << 20
        x = np.sort(O.generate("zipf", n, 3))
        got = ctx.inverse(x)
        assert ctx.timings().factors == n
        want = x[::-1]                                   # n cycles of length 1, smallest index last (unbwts.c:62-86)
        assert np.array_equal(got, want)
        assert np.array_equal(ctx.forward(got), x)


def test_inverse_long_cycle_without_splitter(pkg):
    """B = 1^b 0^c sends i to i + c (mod n); with n = 2^k and c = 2 * odd that is two cycles of n / 2 elements, and the odd one
    meets none of the regular splitters.  Such elements are ranked as nodes of one symbol (memory by their number); the
    earlier answer -- every element a splitter -- needed 129 n bytes."""
    with pkg.Context(0) as ctx:
        for n, c in ((1 << 17, 50002), (1 << 24, 2 * 2777771)):
            B = np.concatenate([np.full(n - c, 1, np.uint8), np.zeros(c, np.uint8)])
            got = ctx.inverse(B)
            t = ctx.timings()
            assert np.array_equal(got, O.inverse(B)), n
            assert t.factors == 2 and t.unvisited == n // 2
        n, c = 1 << 26, 2 * 12345679
        B = np.concatenate([np.full(n - c, 1, np.uint8), np.zeros(c, np.uint8)])
        got = ctx.inverse(B)
        print("device bytes / n at 2^26: %.1f" % (ctx.timings().device_bytes / n))
        assert ctx.timings().device_bytes < 110 * n      # 56 n for the unit nodes (half the elements are unreached) + LF, marks, log, lists
        assert np.array_equal(ctx.forward(got), B)          # bijection: the forward transform of the text is B again
        assert np.array_equal(np.bincount(got, minlength=2), np.bincount(B, minlength=2))


@pytest.mark.parametrize("off_in,off_out", [(1, 0), (3, 5), (0, 7), (13, 2)])
def test_unaligned_device_pointers(ctx, off_in, off_out):
    """Device entry points take any byte address (the vectorised kernels fall back when a pointer is not 16-byte aligned)."""
    n = 300007
    x = O.generate("zipf", n, 17)
    want = O.forward(x)
    d_a, d_b, d_c = ctx.alloc(n + 16), ctx.alloc(n + 16), ctx.alloc(n + 16)
    try:
        d_a.upload(np.concatenate([np.zeros(off_in, dtype=np.uint8), x]))
        ctx.forward_device(d_a.ptr + off_in, n, d_b.ptr + off_out)
        assert np.array_equal(d_b.download()[off_out:off_out + n], want)
        ctx.inverse_device(d_b.ptr + off_out, n, d_c.ptr + off_in)
        assert np.array_equal(d_c.download()[off_in:off_in + n], x)
    finally:
        for b in (d_a, d_b, d_c):
            b.free()


def test_errors(ctx, pkg):
    with pytest.raises(pkg.BwtsError) as e:
        ctx.forward(b"")
    assert e.value.code == -1          # empty input is an error (reference: map_file.c:36-40)
    import ctypes
    assert pkg.lib().bwts_forward_device(ctx._h, None, 10, None) == -1
    h = ctypes.c_void_p()
    assert pkg.lib().bwts_ctx_create(ctypes.byref(h), 1 << 20) == -2


# ---------------------------------------------------------------------------------------------
# BASELINE.json sizes: size-independent properties
# ---------------------------------------------------------------------------------------------
def _properties_at_scale(ctx, kind, n, seed, golden=False):
    d_in, d_out, d_back = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    try:
        ctx.generate(kind, seed, n, d_in)
        ctx.forward_device(d_in, n, d_out)
        tf = ctx.timings()
        # (1) round trip is the identity
        ctx.inverse_device(d_out, n, d_back)
        assert ctx.device_equal(d_in, d_back, n)
        # (2) the transform permutes the input's bytes; (3) bwts[0] = T[n-1]
        x, y = d_in.download(), d_out.download()
        assert np.array_equal(np.bincount(x, minlength=256), np.bincount(y, minlength=256))
        assert y[0] == x[-1]
        # (2b) byte-exact against the oracle where its output at this size is on file (tests/golden/big_forward.json: the pinned
        # CPU restatement of mk_bwts_sa.c:114-195 run in the build container by make_golden_big.py)
        gold = [r for r in BIG["cases"] if r["kind"] == kind and r["n"] == n and r["seed"] == seed]
        if golden:
            assert gold, "no golden for %s(%d, %d)" % (kind, n, seed)
            assert hashlib.sha256(x.tobytes()).hexdigest() == gold[0]["sha256_in"]
            assert hashlib.sha256(y.tobytes()).hexdigest() == gold[0]["sha256_bwts"]
        # (4) the generator stream is what the oracle generates (spot check both ends)
        assert np.array_equal(x[:4096], O.generate(kind, 4096, seed))
        assert np.array_equal(x[-4096:], O.generate(kind, 4096, seed, off=n - 4096))
        # (5) bijection the other way: forward(inverse(x)) == x on the same bytes
        ctx.inverse_device(d_in, n, d_out)
        ctx.forward_device(d_out, n, d_back)
        assert ctx.device_equal(d_in, d_back, n)
        return tf
    finally:
        for b in (d_in, d_out, d_back):
            b.free()


def test_config2_uniform_256MiB_golden_and_properties(ctx):
    """BASELINE config 2, byte-exact: sha-256 of the device's output == the oracle's (golden), then the properties."""
    tf = _properties_at_scale(ctx, "uniform256", 1 << 28, 1, golden=True)
    assert tf.factors >= 1


def test_config3_zipf_1GiB_golden_and_properties(ctx):
    """BASELINE config 3 (the headline workload), byte-exact against the oracle's golden, then the properties."""
    _properties_at_scale(ctx, "zipf", 1 << 30, 1, golden=True)


def test_index_boundary_2p31(ctx):
    """n just above 2^31, beyond the reference's int / saidx_t indices (mk_bwts_sa.c:26-27): the forward bytes are held against
    the golden hash made by the oracle's 64-bit-index instance (tests/golden/big_forward.json, make_golden_big.py; that instance
    is held against the pinned 32-bit one in test_oracle.py), then the size-independent properties."""
    rec = [r for r in BIG["cases"] if r["kind"] == "zipf" and r["n"] == (1 << 31) + 12345][0]
    n = rec["n"]
    d_in, d_out = ctx.alloc(n), ctx.alloc(n)
    try:
        ctx.generate(rec["kind"], rec["seed"], n, d_in)
        assert hashlib.sha256(d_in.download().tobytes()).hexdigest() == rec["sha256_in"]
        ctx.forward_device(d_in, n, d_out)
        assert hashlib.sha256(d_out.download().tobytes()).hexdigest() == rec["sha256_bwts"]
    finally:
        d_in.free()
        d_out.free()
    _properties_at_scale(ctx, rec["kind"], n, rec["seed"])


def test_device_entry_points_take_pinned_host_memory(ctx):
    """bwts_forward_device / bwts_inverse_device with input and output in pinned host memory (bwts_host_alloc): the kernels read the
    text and write the result over PCIe, only the working buffers live on the device -- the route for inputs whose in/out copies do
    not fit beside the working set.  Same bytes as the oracle's."""
    n = (1 << 25) + 777
    x = O.generate("zipf", n, 31)
    want = O.forward(x)
    hin, pin = ctx.host_alloc(n)
    hout, pout = ctx.host_alloc(n)
    try:
        hin[:] = x
        ctx.forward_device(pin, n, pout)
        assert np.array_equal(hout, want)
        hin[:] = want
        hout[:] = 0
        ctx.inverse_device(pin, n, pout)
        assert np.array_equal(hout, x)
    finally:
        ctx.host_free(pin)
        ctx.host_free(pout)


def test_constant_input_at_2p32(ctx):
    """4 GiB of one byte value: n = 2^32 factors of one symbol, every rotation equal -- the one input on which every position stays
    tied at the largest size the 32-bit paths take.  mk_bwts_sa.c:172-188 emits each factor's own last byte: the transform is the
    identity, both ways; small sizes alongside, against the oracle."""
    for m in (1, 7, 100000):
        x = np.full(m, 200, dtype=np.uint8)
        assert np.array_equal(ctx.forward(x), O.forward(x)) and np.array_equal(ctx.inverse(x), x)
    # ... and just beyond 2^32, where the 64-bit paths would refuse it (2^32 + 4097 factor candidates: BWTS_E_RANGE before round 4)
    for n in (1 << 32, (1 << 32) + 4097):
        x = np.full(n, 97, dtype=np.uint8)
        d_in, d_out = ctx.alloc(n), ctx.alloc(n)
        try:
            d_in.upload(x)
            del x
            ctx.forward_device(d_in, n, d_out)
            assert ctx.timings().factors == n
            assert ctx.device_equal(d_in, d_out, n)
            ctx.inverse_device(d_in, n, d_out)
            assert ctx.device_equal(d_in, d_out, n)
        finally:
            d_in.free()
            d_out.free()


def test_config4_dna_4GiB_properties(ctx, pkg):
    """BASELINE config 4: dna(2^32).  Beyond the reference's 32-bit indices (mk_bwts_sa.c:26-27, unbwts.c:12-13), so parity
    is by properties only: round trip both ways, byte histogram preserved, bwts[0] = T[n-1]."""
    try:
        tf = _properties_at_scale(ctx, "dna", 1 << 32, 1)
    except pkg.BwtsError as e:
        if e.code == -3:
            pytest.skip("not enough free device memory for the 4 GiB case")
        raise
    assert tf.key_bits == 2 * tf.key_symbols and tf.key_symbols <= 32      # sigma = 4: 2 bits per symbol


def test_forward_prefix_consistency_64MiB(ctx):
    """Checksum-of-checksums: 64 MiB zipf forward equals the oracle's (sha256 compared)."""
    n = 1 << 26
    x = O.generate("zipf", n, 21)
    assert hashlib.sha256(ctx.forward(x).tobytes()).hexdigest() == hashlib.sha256(O.forward(x).tobytes()).hexdigest()


# ---------------------------------------------------------------------------------------------
# CLI contract end to end (mk_bwts_sa.c:33-65, unbwts.c:19-92,136-176)
# ---------------------------------------------------------------------------------------------
def _wait_gpu_handle_released(pid, timeout=30.0):
    """A finished child's KFD process entry is torn down asynchronously (it held gigabytes of device memory); the GPU
    box allows only a handful of processes on the card at once, so the next child starts only after this one is gone."""
    import time
    entry = "/sys/class/kfd/kfd/proc/%d" % pid
    t0 = time.time()
    while os.path.exists(entry) and time.time() - t0 < timeout:
        time.sleep(0.2)
    time.sleep(0.5)


def _child_report(name, out):
    """What a failed child run said: kept whole under gpurun_out/ (which travels back from the GPU box), head and tail in the
    assertion message -- a child killed by the runtime (a GPU fault aborts the process) says why at the START of its dying words."""
    text = out.decode(errors="replace")
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "child_fail_%s.log" % "".join(c if c.isalnum() else "_" for c in name)[:80]), "w") as f:
            f.write(text)
    except OSError:
        pass
    return text if len(text) <= 6000 else text[:2500] + "\n[...]\n" + text[-3500:]


def _run_cli(args, env=None):
    """subprocess.run for the CLIs (each opens the GPU), followed by the wait above."""
    proc = subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    out, err = proc.communicate(timeout=600)
    _wait_gpu_handle_released(proc.pid)
    return subprocess.CompletedProcess(args, proc.returncode, out, err)


def test_cli_end_to_end(tmp_path):
    x = O.generate("zipf", 300000, 8)
    src = tmp_path / "in.txt"
    x.tofile(src)
    want = O.forward(x).tobytes()
    r = _run_cli([os.path.join(PKG, "mk_bwts"), str(src)])
    assert r.returncode == 0 and r.stdout == want                      # no outfile -> stdout
    out = tmp_path / "out.bwts"
    r = _run_cli([os.path.join(PKG, "mk_bwts"), str(src), str(out)])
    assert r.returncode == 0 and r.stdout == b"" and out.read_bytes() == want
    back = tmp_path / "back.txt"
    r = _run_cli([os.path.join(PKG, "unbwts"), str(out), str(back)])
    assert r.returncode == 0 and back.read_bytes() == x.tobytes()
    r = _run_cli([os.path.join(PKG, "unbwts"), str(out)])
    assert r.returncode == 0
    line = r.stdout.decode().strip()
    assert line.startswith("Writing to %s_" % out)                      # unbwts.c:152-158
    assert open(line[len("Writing to "):], "rb").read() == x.tobytes()
    r = _run_cli([os.path.join(PKG, "mk_bwts"), str(src), str(tmp_path / "nodir" / "x")])
    assert r.returncode == 1 and r.stderr.decode().startswith("Couldn't open BWTS file for writing")   # mk_bwts_sa.c:55-59
    env = dict(os.environ, BWTS_TIMINGS="1")
    r = _run_cli([os.path.join(PKG, "mk_bwts"), str(src), str(out)], env=env)
    labels = [l.split(" time ")[0] for l in r.stderr.decode().splitlines() if " time " in l]
    assert labels[:5] == ["Suffix sort", "Compute ISA", "Fix sort order", "Generate BWTS", "Write BWTS"]   # mk_bwts_sa.c:50,124,168,190,62
    assert out.read_bytes() == want
    # the variant program the reference's `make test` drives: auto-named output (mk_bwts_new_algo.c:210-216)
    r = _run_cli([os.path.join(PKG, "mk_bwts_new_algo"), str(src)])
    assert r.returncode == 0
    line = r.stdout.decode().strip()
    assert line.startswith("Writing to %s_" % src) and line.endswith(".bwts")
    assert open(line[len("Writing to "):], "rb").read() == want
    r = _run_cli([os.path.join(PKG, "mk_bwts_new_algo")])
    assert r.returncode == 1 and r.stderr.decode().splitlines()[1] == "If outfile is not supplied, a unique file name is generated"


def test_cli_in_place_and_failure_leaves_nothing(tmp_path):
    """The output is opened when the first piece arrives (the reference transforms first and opens afterwards, mk_bwts_sa.c:47-60,
    unbwts.c:62-89): a file can be transformed onto itself, and a transform that fails neither truncates an existing destination
    nor leaves an auto-named file behind."""
    x = O.generate("zipf", 300007, 23)
    want = O.forward(x).tobytes()
    f = tmp_path / "same.bin"
    f.write_bytes(x.tobytes())
    r = _run_cli([os.path.join(PKG, "mk_bwts"), str(f), str(f)])
    assert r.returncode == 0 and f.read_bytes() == want
    r = _run_cli([os.path.join(PKG, "unbwts"), str(f), str(f)])
    assert r.returncode == 0 and f.read_bytes() == x.tobytes()
    # a forced failure: the blocked 64-bit path without fallback refuses a^n (BWTS_E_RANGE)
    a_n = tmp_path / "an.bin"
    a_n.write_bytes(b"a" * 100000)
    dest = tmp_path / "precious.bwts"
    dest.write_bytes(b"keep me")
    env = dict(os.environ, BWTS_TEST_KNOBS="1", BWTS_FORCE_WIDE="2")
    r = _run_cli([os.path.join(PKG, "mk_bwts"), str(a_n), str(dest)], env=env)
    assert r.returncode == 1 and b"transform failed" in r.stderr and dest.read_bytes() == b"keep me"
    before = set(os.listdir(tmp_path))
    r = _run_cli([os.path.join(PKG, "mk_bwts_new_algo"), str(a_n)], env=env)
    assert r.returncode == 1 and set(os.listdir(tmp_path)) == before and b"Writing to" not in r.stdout


def test_make_test_prints_match():
    """The reference's golden-file check (Makefile:30-33): mk_bwts_new_algo on testdata/testjunk, cmp, "Match"."""
    proc = subprocess.Popen(["make", "-C", PKG, "test"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out, _ = proc.communicate(timeout=600)
    time.sleep(1.0)
    assert proc.returncode == 0, out.decode()[-2000:]
    assert out.decode().strip().splitlines()[-1] == "Match" or "Match" in out.decode().splitlines()


# ---------------------------------------------------------------------------------------------
# host-buffer path (what the CLIs use): staging ring, copy workers, sink, pinned blocks
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 4095, (8 << 20) - 1, 8 << 20, (8 << 20) + 1, (40 << 20) + 12345])
def test_host_path_sizes(ctx, n):
    """Sizes around the staging chunk (8 MiB) and beyond the ring (4 slots): same bytes as the device path's oracle check."""
    x = O.generate("zipf", n, 31)
    want = O.forward(x) if n <= (8 << 20) + 1 else None
    y = ctx.forward(x)
    if want is not None:
        assert np.array_equal(y, want)
    assert np.array_equal(ctx.forward_sink(x), y)                 # same result through the sink callback
    assert np.array_equal(ctx.inverse(y), x)
    assert np.array_equal(ctx.inverse_sink(y), x)


def test_host_path_pinned_blocks(ctx):
    n = (9 << 20) + 77
    x = O.generate("text", n, 4)
    a, pa = ctx.host_alloc(n)
    b, pb = ctx.host_alloc(n)
    try:
        a[:] = x
        ctx.forward_into(a, b)                                    # both pinned: one DMA each way
        y = b.copy()
        assert np.array_equal(y, O.forward(x))
        plain = np.empty(n, dtype=np.uint8)
        ctx.forward_into(a, plain)                                # pinned in, pageable out
        assert np.array_equal(plain, y)
        ctx.inverse_into(y, a)                                    # pageable in, pinned out
        assert np.array_equal(a, x)
    finally:
        del a, b
        ctx.host_free(pa)
        ctx.host_free(pb)


def test_batch_equals_single_calls(ctx, pkg):
    """bwts_forward_batch / bwts_inverse_batch (copies of the neighbouring items overlapped with the transform): the bytes of the
    single calls, for items of very different sizes; a bad item is refused before anything runs."""
    xs = [O.generate("zipf", (3 << 20) + 1, 31), np.frombuffer(b"b", dtype=np.uint8), O.generate("uniform256", (9 << 20) + 5, 32),
          O.generate("text", 5 << 20, 33), np.frombuffer(b"mississippi banana", dtype=np.uint8), O.generate("dna", 1 << 20, 34),
          O.generate("zipf", 17 << 20, 35)]
    ys = ctx.forward_batch(xs)
    for x, y in zip(xs, ys):
        assert np.array_equal(y, O.forward(x))
    backs = ctx.inverse_batch(ys)
    for x, b in zip(xs, backs):
        assert np.array_equal(b, x)
    assert ctx.forward_batch([]) == []
    import ctypes
    bad_in = (ctypes.c_void_p * 2)(xs[0].ctypes.data, None)
    outs = [np.empty_like(xs[0]), np.empty_like(xs[0])]
    bad_out = (ctypes.c_void_p * 2)(outs[0].ctypes.data, outs[1].ctypes.data)
    ns = (ctypes.c_uint64 * 2)(xs[0].size, 5)
    assert pkg.lib().bwts_forward_batch(ctx._h, 2, bad_in, ns, bad_out) == -1
    ns0 = (ctypes.c_uint64 * 2)(xs[0].size, 0)
    ok_in = (ctypes.c_void_p * 2)(xs[0].ctypes.data, xs[0].ctypes.data)
    assert pkg.lib().bwts_forward_batch(ctx._h, 2, ok_in, ns0, bad_out) == -1


def test_arena_regrown_on_a_reused_context(pkg):
    """a context that already holds an arena meets a larger input through the host-buffer entry point.  The old
    block is given up on the calling thread with the stream drained, before the input is staged; growth of 256 MiB and more used
    to be done on a helper thread beside the copies.  Fresh context (its first reservation is the one case that still overlaps the
    hipMalloc with the input copy), small call, then 12 MiB (arena + 300 MiB), then small again."""
    small = O.generate("uniform256", 70001, 5)
    big = O.generate("zipf", 12 << 20, 6)
    with pkg.Context(0) as c:
        assert np.array_equal(c.forward(small), O.forward(small))
        held = c.timings().device_bytes
        y = c.forward(big)
        assert c.timings().device_bytes >= held + (256 << 20)
        assert np.array_equal(y, O.forward(big))
        assert np.array_equal(c.inverse(y), big)
        assert np.array_equal(c.forward(small), O.forward(small))
    with pkg.Context(0) as c:                                       # first call of a context: the helper's case
        assert np.array_equal(c.forward(big), y)


def test_failed_call_leaves_the_output_buffer_alone(ctx, pkg):
    """The output pages are pre-faulted while the GPU works without changing their contents (MADV_POPULATE_WRITE), so a call that
    fails after that point has not written to the caller's buffer (the reference writes only after success: mk_bwts_sa.c:52-60).
    Forced here with a sink-less call whose transform is refused: length 0 is refused before anything; a real late failure is not
    constructible on demand, so the property is checked where it is observable -- in place, output == input buffer, >= 4 MiB:
    a pre-fault that wrote zeros would corrupt every 4096th input byte before staging had read it."""
    x = O.generate("text", (6 << 20) + 3, 9)
    want = ctx.forward(x)
    buf = x.copy()
    ctx.forward_into(buf, buf)
    assert np.array_equal(buf, want)
    ctx.inverse_into(buf, buf)
    assert np.array_equal(buf, x)


def test_batch_in_place_and_overlap_rules(ctx, pkg):
    """Batch items may be transformed in place (outs[k] == ins[k], items beyond the 4 MiB pre-fault threshold); an output that
    overlaps a LATER item's input is refused (bwts.h)."""
    import ctypes
    xs = [O.generate("zipf", (5 << 20) + 7, 61), O.generate("text", 6 << 20, 62), O.generate("uniform256", (4 << 20) + 4096, 63)]
    want = [ctx.forward(x) for x in xs]
    bufs = [x.copy() for x in xs]
    k = len(bufs)
    ptrs = (ctypes.c_void_p * k)(*[b.ctypes.data for b in bufs])
    ns = (ctypes.c_uint64 * k)(*[b.size for b in bufs])
    assert pkg.lib().bwts_forward_batch(ctx._h, k, ptrs, ns, ptrs) == 0
    for b, w in zip(bufs, want):
        assert np.array_equal(b, w)
    assert pkg.lib().bwts_inverse_batch(ctx._h, k, ptrs, ns, ptrs) == 0
    for b, x in zip(bufs, xs):
        assert np.array_equal(b, x)
    # output 0 on input 1 (a later item): refused, nothing changed
    outs = (ctypes.c_void_p * k)(bufs[1].ctypes.data, bufs[0].ctypes.data, bufs[2].ctypes.data)
    ns2 = (ctypes.c_uint64 * k)(min(bufs[0].size, bufs[1].size), min(bufs[0].size, bufs[1].size), bufs[2].size)
    assert pkg.lib().bwts_forward_batch(ctx._h, k, ptrs, ns2, outs) == -1
    for b, x in zip(bufs, xs):
        assert np.array_equal(b, x)


def test_sink_error_aborts(ctx, pkg):
    import ctypes
    x = O.generate("zipf", 100000, 2)
    cb = pkg.SINK_FN(lambda user, ptr, length: 1)
    assert pkg.lib().bwts_forward_sink(ctx._h, x.ctypes.data, x.size, cb, None) == -7
    assert np.array_equal(ctx.forward(x), O.forward(x))           # the context is still usable


def test_two_contexts_one_process(pkg):
    """Contexts are independent (bwts.h): two on device 0 back to back and interleaved -- and one per device when the box has
    more than one.  Every context opts its own kernels into > 64 KB of LDS."""
    import torch
    x = O.generate("zipf", 1 << 20, 41)                           # large enough for the packed round-0 passes
    z = O.generate("text", 1 << 20, 42)                           # dense ranks: the binned rank build's large-LDS kernels
    wx, wz = O.forward(x), O.forward(z)
    devs = [0, 0] + ([1] if torch.cuda.device_count() > 1 else [])
    ctxs = [pkg.Context(d) for d in devs]
    try:
        for c in ctxs:
            assert np.array_equal(c.forward(x), wx)
        for c in reversed(ctxs):
            assert np.array_equal(c.forward(z), wz)
            assert np.array_equal(c.inverse(wz), z)
    finally:
        for c in ctxs:
            c.close()


def test_timing_is_opt_in(ctx):
    x = O.generate("zipf", 1 << 20, 5)
    ctx.set_timing(0)
    ctx.forward(x)
    t = ctx.timings().as_dict()
    assert t["total_ms"] > 0 and t["kernels"]["radix_scatter"]["launches"] > 0 and t["kernels"]["radix_scatter"]["ms"] == 0
    ctx.set_timing(2)
    ctx.forward(x)
    t = ctx.timings().as_dict()
    assert t["kernels"]["radix_scatter"]["ms"] > 0
    ctx.set_timing(0)


# ---------------------------------------------------------------------------------------------
# beyond 32-bit indices: the blocked forward path with 64-bit positions and ranks
# ---------------------------------------------------------------------------------------------
WIDE_CASES = [("zipf", 70001, 3), ("dna", 300007, 4), ("uniform256", 65536, 5), ("zipf", (1 << 20) + 17, 6), ("text", 200003, 7),
              ("dna", (1 << 21) + 3, 8)]


def test_wide_path_small_vs_oracle_child():
    """The n > 2^32 path forced onto small inputs (BWTS_FORCE_WIDE=2: no fallback), with segments of 2^13 positions and buckets of
    n/6 elements so that many segments and buckets take part: bytes equal the oracle's, and the main path inverts them."""
    if os.environ.get("BWTS_TEST_CHILD"):
        pytest.skip("already inside a child run")
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle_lib as O, __graft_entry__ as ge
import os
pkg = ge.load_package()
for kind, n, seed in %r:
    os.environ["BWTS_WIDE_BUCKET"] = str(max(256, n // 6))
    x = O.generate(kind, n, seed)
    with pkg.Context(0) as ctx:
        y = ctx.forward(x)
        t = ctx.timings()
        assert np.array_equal(y, O.forward(x)), (kind, n)
        assert np.array_equal(ctx.inverse(y), x)
        print(kind, n, "rounds", t.rounds, "tied", t.active_after_round0, "factors", t.factors)
# the rounds take the tied list in parts of whole groups and the list lives in blocks: small parts and blocks, so that a round has
# dozens of parts, survivors cross block borders, and later parts read ranks that earlier parts of the same round refined
for kind, n, seed, part, lg in (("text", 300007, 11, 3000, 12), ("text", (1 << 20) + 5, 12, 20000, 14), ("dna", 200003, 13, 64, 8),
                                ("zipf", 400009, 14, 1000, 10)):
    os.environ["BWTS_WIDE_BUCKET"] = str(max(256, n // 5))
    os.environ["BWTS_WIDE_PART"] = str(part); os.environ["BWTS_WIDE_TBLOCK_LOG2"] = str(lg)
    os.environ["BWTS_WIDE_DIRECT"] = "0"               # (the rank array and the rounds at once, also where the ties are few)
    x = O.generate(kind, n, seed)
    with pkg.Context(0) as ctx:
        y = ctx.forward(x)
        t = ctx.timings()
        assert np.array_equal(y, O.forward(x)), ("parts", kind, n)
        print("parts:", kind, n, "rounds", t.rounds, "tied", t.active_after_round0, "parts of", part, "blocks of", 1 << lg)
        if kind == "text": assert t.active_after_round0 > 8 * part
del os.environ["BWTS_WIDE_PART"], os.environ["BWTS_WIDE_TBLOCK_LOG2"]
# few ties: no rank array at all, the tied groups ordered by comparing rotations in the text (forced: a fallback would be an error);
# text needs the ranks and says so
os.environ["BWTS_WIDE_DIRECT"] = "1"
for kind, n, seed in (("zipf", 250001, 21), ("dna", 400003, 22), ("uniform256", 100000, 23), ("dna", (1 << 21) + 11, 24)):
    os.environ["BWTS_WIDE_BUCKET"] = str(max(256, n // 7))
    x = O.generate(kind, n, seed)
    with pkg.Context(0) as ctx:
        y = ctx.forward(x)
        t = ctx.timings()
        assert np.array_equal(y, O.forward(x)), ("direct", kind, n)
        print("direct:", kind, n, "tied", t.active_after_round0, "factors", t.factors)
x = O.generate("text", 200003, 25)
with pkg.Context(0) as ctx:
    try:
        ctx.forward(x)
        raise AssertionError("text went through without the rank array")
    except pkg.BwtsError as e:
        assert e.code == -5, e
del os.environ["BWTS_WIDE_DIRECT"]
for kat in (b"banana", b"mississippi", b"abracadabra", b"the quick brown fox jumps over the lazy dog"):
    os.environ["BWTS_WIDE_BUCKET"] = "256"
    with pkg.Context(0) as ctx:
        assert ctx.forward(kat).tobytes() == O.forward(kat).tobytes(), kat
# one context for a run of inputs (arenas and registers carry over), among them sorted runs: thousands of one-symbol Lyndon
# factors, i.e. thousands of single-lane waves in the key patch kernel (the case behind tools/check_shift64.py)
os.environ["BWTS_WIDE_BUCKET"] = "4096"
rng = np.random.default_rng(9038)
with pkg.Context(0) as ctx:
    for rep in range(6):
        big = O.generate("zipf", 150000 + rep, 40 + rep)
        assert np.array_equal(ctx.forward(big), O.forward(big)), ("shared context, zipf", rep)
        n = int(rng.integers(5000, 12000))
        x = np.sort(rng.integers(0, 100, size=n, dtype=np.uint8))[::-1].copy()
        c = int(rng.integers(0, n))
        x = (np.concatenate([x[c:], x[:c]]) + 48).astype(np.uint8)
        y = ctx.forward(x)
        assert np.array_equal(y, O.forward(x)), ("shared context, sorted run", rep, n)
        assert np.array_equal(ctx.inverse(y), x)
# inverse of data whose LF map has a long cycle that meets no regular splitter: B = 1^b 0^c sends i to i + c (mod n); with
# n = 2^k and c = 2 * odd that is two cycles of n / 2 elements, the odd one free of multiples of 256 (unit-node ranking)
with pkg.Context(0) as ctx:
    for n, c in ((1 << 17, 50002), (1 << 20, 2 * 177771)):
        B = np.concatenate([np.full(n - c, 1, np.uint8), np.zeros(c, np.uint8)])
        assert np.array_equal(ctx.inverse(B), O.inverse(B)), ("long cycle without a splitter", n)
        x = O.generate("zipf", 100000, n)
        assert np.array_equal(ctx.inverse(x), O.inverse(x))
print("wide ok")
""" % (ROOT, os.path.join(ROOT, "tests"), WIDE_CASES)
    env = dict(os.environ, BWTS_TEST_CHILD="1", BWTS_TEST_KNOBS="1", BWTS_FORCE_WIDE="2", BWTS_WIDE_SEG_LOG2="13")
    proc = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, cwd=ROOT)
    out, _ = proc.communicate(timeout=900)
    _wait_gpu_handle_released(proc.pid)
    assert proc.returncode == 0 and b"wide ok" in out, _child_report("wide", out)


def _lf_walk_matches_text(x, y, steps):
    """Follows LF from slot 0 of y for `steps` steps (unbwts.c:50-52, 66-82 on the host, rank by block counts): the bytes met must
    be the text read backwards from its end, for as long as the walk stays inside the last Lyndon factor's cycle."""
    n = y.size
    B = 1 << 20
    nb = (n + B - 1) // B
    syms = np.flatnonzero(np.bincount(y[: 1 << 24], minlength=256) + np.bincount(y[-(1 << 24):], minlength=256))
    total = np.bincount(y, minlength=256) if n <= (1 << 32) else None
    cum = {}
    counts = np.zeros(256, dtype=np.int64)
    for c in syms:
        per = np.zeros(nb + 1, dtype=np.int64)
        for b in range(nb):
            per[b + 1] = per[b] + int(np.count_nonzero(y[b * B:(b + 1) * B] == c))
        cum[int(c)] = per
        counts[c] = per[-1]
    assert counts.sum() == n                                   # every byte value of y was seen in the two sampled ends
    C = np.concatenate([[0], np.cumsum(counts)])[:256]
    r = 0
    for t in range(steps):
        c = int(y[r])
        if c != int(x[n - 1 - t]):
            return t
        b = r // B
        r = int(C[c] + cum[c][b] + np.count_nonzero(y[b * B:r] == c))
        if r == 0:
            return steps                                       # the cycle closed: the whole last factor was reproduced
    return steps


def test_wide_12GiB_dna_round_trip(ctx, pkg):
    """dna(12 GiB): beyond 2^32 positions, forward and inverse through the 64-bit paths.  Round trip exact (compared on the host);
    the bytes are a permutation of the input's, bwts[0] = T[n-1] (mk_bwts_sa.c:188), and 200 000 steps of the inverse's LF walk
    from slot 0 -- done on the host with exact ranks, independently of the engine's inverse -- read the text backwards from its
    end (a sampled check of the forward bytes: each step tests one slot)."""
    n = 12 << 30
    try:
        d_in, d_out = ctx.alloc(n), ctx.alloc(n)
    except pkg.BwtsError:
        pytest.skip("not enough device memory")
    d_back = None
    try:
        ctx.generate("dna", 1, n, d_in)
        try:
            ctx.forward_device(d_in, n, d_out)
        except pkg.BwtsError as e:
            if e.code == -3:
                pytest.skip("not enough free device memory for the 12 GiB case")
            raise
        t = ctx.timings()
        x = d_in.download()
        d_in.free()
        y = d_out.download()
        d_back = ctx.alloc(n)
        ctx.inverse_device(d_out, n, d_back)
        ti = ctx.timings()
        back = d_back.download()
    finally:
        for b in (d_in, d_out, d_back):
            if b is not None:
                b.free()
    assert np.array_equal(back, x)                               # unbwts o mk_bwts = id at 12 GiB
    del back
    assert y[0] == x[-1]
    assert np.array_equal(O.generate("dna", 4096, 1, off=n - 4096), x[-4096:])
    hx = sum(np.bincount(x[i:i + (1 << 30)], minlength=256) for i in range(0, n, 1 << 30))
    hy = sum(np.bincount(y[i:i + (1 << 30)], minlength=256) for i in range(0, n, 1 << 30))
    assert np.array_equal(hx, hy)
    assert _lf_walk_matches_text(x, y, 200000) == 200000
    assert t.factors >= 1 and t.rounds >= 1 and ti.factors == t.factors     # LF cycles = Lyndon factors
    print("12 GiB: forward %.0f ms, inverse %.0f ms, factors %d" % (t.total_ms, ti.total_ms, t.factors))


def test_wide_6GiB_text_round_trip(ctx, pkg):
    """text(6 GiB): beyond 2^32 positions with 71 % of them tied after the first sort (4.6 * 10^9): the tied list in blocks taken as it
    grows, the rounds over it part by part (three parts per round at first), 64-bit ranks.  A sample of the keys tells the forward that this input
    needs the rank array (i.i.d. data runs without).  Round trip exact on the device, the bytes a permutation of
    the input's, bwts[0] = T[n-1]."""
    n = 6 << 30
    ctx.release_memory()                     # (what earlier tests left with the context: this case takes 215 GiB of its own)
    try:
        d_in, d_out, d_back = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    except pkg.BwtsError:
        pytest.skip("not enough device memory")
    try:
        ctx.generate("text", 1, n, d_in)
        try:
            ctx.forward_device(d_in, n, d_out)
        except pkg.BwtsError as e:
            if e.code == -3:
                pytest.skip("not enough free device memory for the 6 GiB text case")
            raise
        t = ctx.timings()
        assert t.active_after_round0 > n // 2 and t.rounds >= 10
        ctx.inverse_device(d_out, n, d_back)
        assert ctx.device_equal(d_in, d_back, n)                 # unbwts o mk_bwts = id
        x, y = d_in.download(), d_out.download()
    finally:
        for b in (d_in, d_out, d_back):
            b.free()
    assert y[0] == x[-1]
    assert np.array_equal(O.generate("text", 4096, 1, off=n - 4096), x[-4096:])
    hx = sum(np.bincount(x[i:i + (1 << 30)], minlength=256) for i in range(0, n, 1 << 30))
    hy = sum(np.bincount(y[i:i + (1 << 30)], minlength=256) for i in range(0, n, 1 << 30))
    assert np.array_equal(hx, hy)
    print("6 GiB text: forward %.0f ms, tied %d, rounds %d, factors %d" % (t.total_ms, t.active_after_round0, t.rounds, t.factors))
    ctx.release_memory()


def test_lf_walk_checker_on_small_input():
    """The host-side checker used at 12 GiB, held against a case the oracle covers.  It is a sampled check: every step tests one
    slot's byte with exact ranks, so damage has to touch a fair share of the slots (here 5 %) to be met within the walk."""
    x = O.generate("dna", (1 << 21) + 5, 3)
    y = O.forward(x)
    assert _lf_walk_matches_text(x, y, 50000) == 50000
    z = y.copy()
    lo, hi = z.size // 2, z.size // 2 + z.size // 20
    z[lo:hi] = z[lo:hi][::-1].copy()
    assert _lf_walk_matches_text(x, z, 50000) < 50000


def test_wide_path_1GiB_equals_main_path_child():
    """zipf(2^30) through the n > 2^32 path (forced; segments of 2^27 positions, buckets of 2^27 elements) gives byte for byte
    what the main path gives: the blocked collection, the per-bucket sorts and the 64-bit rounds at a real size."""
    if os.environ.get("BWTS_TEST_CHILD"):
        pytest.skip("already inside a child run")
    code = r"""
import sys, os
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package()
n = 1 << 30
with pkg.Context(0) as ctx:
    a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    ctx.generate("zipf", 1, n, a)
    ctx.forward_device(a, n, b)                     # forced wide (BWTS_FORCE_WIDE=2 in the environment)
    t = ctx.timings()
    print("wide: %%.1f ms, rounds %%d, tied %%d, factors %%d" %% (t.total_ms, t.rounds, t.active_after_round0, t.factors))
    ctx.inverse_device(b, n, c)
    assert ctx.device_equal(a, c, n)                # the main inverse turns it back into the input
    # text(2^30): 750 M positions tied after round 0 -- the tied list in blocks, every round in three parts -- against the oracle's
    # bytes at this size (tests/golden/big_forward.json)
    import hashlib, json
    gold = [r for r in json.load(open(%r))["cases"] if r["kind"] == "text" and r["n"] == n and r["seed"] == 1][0]
    ctx.generate("text", 1, n, a)
    ctx.forward_device(a, n, b)
    t = ctx.timings()
    print("wide text: %%.1f ms, rounds %%d, tied %%d, factors %%d" %% (t.total_ms, t.rounds, t.active_after_round0, t.factors))
    assert t.active_after_round0 > n // 2
    assert hashlib.sha256(b.download().tobytes()).hexdigest() == gold["sha256_bwts"]
print("wide ok")
""" % (ROOT, os.path.join(ROOT, "tests", "golden", "big_forward.json"))
    env = dict(os.environ, BWTS_TEST_CHILD="1", BWTS_TEST_KNOBS="1", BWTS_FORCE_WIDE="2", BWTS_WIDE_SEG_LOG2="27", BWTS_WIDE_BUCKET=str(1 << 27))
    proc = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, cwd=ROOT)
    out, _ = proc.communicate(timeout=900)
    _wait_gpu_handle_released(proc.pid)
    assert proc.returncode == 0 and b"wide ok" in out, _child_report("wide", out)


# ---------------------------------------------------------------------------------------------
# repeat-rich text (the reference's named workload is enwik8, Makefile:35-38)
# ---------------------------------------------------------------------------------------------
def test_text_generator_matches_oracle(ctx):
    n = (1 << 22) + 3
    d = ctx.alloc(n)
    try:
        ctx.generate("text", 9, n, d)
        assert hashlib.sha256(d.download().tobytes()).hexdigest() == hashlib.sha256(O.generate("text", n, 9).tobytes()).hexdigest()
    finally:
        d.free()


def test_text_16MiB_vs_oracle(ctx):
    """16 MiB of the text workload: most positions tied after round 0, many rounds on the tied list."""
    n = 1 << 24
    x = O.generate("text", n, 1)
    y = ctx.forward(x)
    t = ctx.timings()
    assert t.active_after_round0 > n // 4 and t.rounds >= 3
    assert hashlib.sha256(y.tobytes()).hexdigest() == hashlib.sha256(O.forward(x).tobytes()).hexdigest()
    assert np.array_equal(ctx.inverse(y), x)


def test_real_text_vs_oracle(ctx):
    """Real text with long repeats -- source and documentation files of this image, 53.6 MiB, 205 distinct bytes, 82 % of the positions
    tied after round 0, groups of thousands of members: the device's bytes against the oracle run here on the same bytes and
    against its golden hash (tests/golden/realtext.json); skipped where the image's files differ from the golden's."""
    import realtext
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "realtext.json")))
    x = np.frombuffer(realtext.corpus(1 << gold["limit_log2"]), dtype=np.uint8)
    if x.size != gold["n"] or hashlib.sha256(x.tobytes()).hexdigest() != gold["sha256_in"]:
        pytest.skip("this box's file set is not the golden's")
    y = ctx.forward(x)
    t = ctx.timings()
    assert t.active_after_round0 > x.size // 2 and t.rounds >= 3
    assert hashlib.sha256(y.tobytes()).hexdigest() == gold["sha256_bwts"]
    assert np.array_equal(y, O.forward(x))
    assert np.array_equal(ctx.inverse(y), x)


def test_real_text_1GiB_golden(ctx):
    """1 GiB of REAL text -- Python packages, ROCm headers and data files, source and documentation of this image (tests/realtext.py
    corpus_big) -- the size BASELINE's metric is quoted on: the device's bytes against the oracle's golden hash
    (tests/golden/realtext_1GiB.json, made by make_golden_realtext_big.py: the oracle needs ten minutes for it), then the round trip.
    Skipped where the image's files differ from the golden's."""
    import realtext
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "realtext_1GiB.json")))
    x = np.frombuffer(realtext.corpus_big(1 << 30), dtype=np.uint8)
    if x.size != gold["n"] or hashlib.sha256(x.tobytes()).hexdigest() != gold["sha256_in"]:
        pytest.skip("this box's file set is not the golden's")
    y = ctx.forward(x)
    t = ctx.timings()
    assert t.active_after_round0 > x.size // 2
    assert hashlib.sha256(y.tobytes()).hexdigest() == gold["sha256_bwts"]
    assert np.array_equal(ctx.inverse(y), x)


def test_text_1GiB_golden_and_properties(ctx):
    """The bench's text workload at full size, byte-exact against the oracle's golden, then the properties."""
    _properties_at_scale(ctx, "text", 1 << 30, 1, golden=True)


# ---------------------------------------------------------------------------------------------
# alternate code paths, selected by environment knobs (read once per context, and only under BWTS_TEST_KNOBS=1 -> one child process per mode)
# ---------------------------------------------------------------------------------------------
def test_stray_knobs_are_ignored_without_the_gate(pkg):
    """A process that merely inherits BWTS_* variables runs the product paths: the switches for alternate code paths count only
    under BWTS_TEST_KNOBS=1, and a context reads its environment once, when it is made."""
    x = O.generate("zipf", 200000, 77)
    os.environ["BWTS_FORCE_WIDE"] = "2"
    os.environ["BWTS_KEY_BITS"] = "16"
    try:
        with pkg.Context(0) as c:
            y = c.forward(x)
            t = c.timings()
            assert t.key_bits > 16                             # not the forced width: the knob was not looked at
            assert np.array_equal(y, O.forward(x))
        os.environ["BWTS_TEST_KNOBS"] = "1"
        del os.environ["BWTS_FORCE_WIDE"]
        with pkg.Context(0) as c:
            y = c.forward(x)
            assert c.timings().key_bits == 16 and np.array_equal(y, O.forward(x))
    finally:
        for k in ("BWTS_FORCE_WIDE", "BWTS_KEY_BITS", "BWTS_TEST_KNOBS"):
            os.environ.pop(k, None)


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


# (last in the file: should one of these children die -- DESIGN.md section 10 -- every other test of the suite has run by then)
_ALT_ENVS = [
    {"BWTS_VARLEN": "1", "BWTS_KEY_BITS": "24"},      # variable-length key codes on every input, narrow keys: many ties, sparse ranks
    {"BWTS_VARLEN": "0", "BWTS_KEY_SYMBOLS": "2"},    # fixed-width keys of two symbols: nearly everything tied, dense ranks
    {"BWTS_LYNDON": "general"},                       # factors from a full suffix sort + prefix minima of ISA
    {"BWTS_EMIT": "gather"},                          # classic bwts[r] = P[sa[r]] gather instead of the carried byte
    {"BWTS_RX_PACK": "0"},                            # round-0 sort on wide (u64, u32, u8) streams instead of packed ones
    {"BWTS_GROUPSCAN": "keys"},                       # round-0 group scan element-wise over the keys instead of flag words
    {"BWTS_RANKBUILD": "plain"},
    {"BWTS_DENSE": "tiles", "BWTS_DENSE_RUNS": "1"},  # (tile form) activation rounds from the run structure of the position-ordered list (opt-in)
    {"BWTS_DENSE_STEP": "2"},                         # group-local rounds with plain doubling (one successor rank) instead of the quadrupled step
    {"BWTS_DENSE": "tiles"},                          # the tile form of the group-local rounds (round 2) instead of the chunked one
    {"BWTS_RX_SMALL": "0"},                           # small sorts through the multi-launch passes instead of the one-workgroup kernel
    {"BWTS_RX_CHAIN": "1"},                           # round-0 packed passes with decoupled look-back instead of histogram sweep + column scan (>= 2^22 elements; opt-in: measured slower)
    {"BWTS_RX_FUSED_SCAN": "0"},                      # column scan of the tile table in five launches instead of the fused kernel
    {"BWTS_K0DIR": "0"},                              # sparse key builder: plain binary searches, no directories                      # dense rank array by two plain scatters instead of the binned one
    {"BWTS_INV_MARK": "log"},                         # inverse logs every visited index (the fallback of the per-range moments)
    {"BWTS_INV_MARK": "sentinel"},                    # inverse marks visited entries in place instead of logging them
    {"BWTS_BYTEMARK": "1"},                           # inverse marks in a byte map (the n = 2^32 fallback)
    {"BWTS_SPLIT_LOG2": "0"},                         # inverse: every element a splitter (plain pointer jumping)
    {"BWTS_POISON": "1"},                             # every arena / side block filled with 0xA5 before use: nothing may read what nothing wrote
    {"BWTS_PARK": "1"},                               # later rounds with PARKED CHAINS (csrc/chunk_rounds.h; opt-in: measured slower than the default rounds)
    {"BWTS_PARK": "1", "BWTS_PARK_STATIC": "1"},      # ... and long repeats parked before the first round from the sorted group records
    {"BWTS_RESERVE_HELPER": "1"},                     # host path: EVERY arena growth through the helper thread (release on the caller with the stream drained, hipMalloc alone on the helper: DESIGN.md section 9)
]
_alt_pool = {}


def _alt_child(env):
    # (-s: the child must not capture its tests' stderr -- what the HIP runtime says when it aborts the process would stay in the capture file)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-s",
           "-k", "(small or mid_size or deep_repeats or dense_ties or dense_rounds or chunk_rounds or text_16MiB or reference_unbwts_vectors_through_cabi) and not alternate"]
    # (-k matches case-insensitively and looks at parameter ids too: without the exclusion, an id like BWTS_RX_SMALL=0
    # makes the child select this very test and start a child of its own.)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                            env=dict(os.environ, BWTS_TEST_CHILD="1", BWTS_TEST_KNOBS="1", LIBC_FATAL_STDERR_="1", BWTS_TRACE_ALLOC="1", **env), cwd=ROOT)
    try:
        out, _ = proc.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
    _wait_gpu_handle_released(proc.pid)
    return proc.returncode, out


@pytest.mark.parametrize("env", _ALT_ENVS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_alternate_paths(env):
    """One child process per mode (a context reads its knobs once), THREE at a time: with this process that is four users of the
    card, inside what a GPU box allows; the first of these tests starts them all, each test then waits for its own child."""
    if os.environ.get("BWTS_TEST_CHILD"):
        pytest.skip("already inside a child run")
    if not _alt_pool:
        from concurrent.futures import ThreadPoolExecutor
        ex = ThreadPoolExecutor(max_workers=3)
        for e in _ALT_ENVS:
            _alt_pool[tuple(sorted(e.items()))] = ex.submit(_alt_child, e)
    rc, out = _alt_pool[tuple(sorted(env.items()))].result()
    assert rc == 0, _child_report("alternate_" + ",".join("%s=%s" % kv for kv in env.items()), out)
