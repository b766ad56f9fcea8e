/*
 * bwts_oracle.h -- CPU oracle for the bijective BWT (BWTS) hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 * The product (libbwts_hip.so, mk_bwts, unbwts) never links or calls it.
 *
 * Parity status: PINNED.
 *   - forward: every known-answer vector and large-input hash recorded in
 *     SURVEY.md 8(c) (produced by the compiled reference) is checked in
 *     tests/test_oracle.py; additionally oracle_forward() output is fed to the
 *     real reference inverse (oracle/_ref/unbwts, built from
 *     /root/reference/unbwts.c + map_file.c, unmodified) which must return the
 *     input -- the transform is a bijection, so that pins forward bytes exactly.
 *   - inverse: compared byte-for-byte with oracle/_ref/unbwts.
 *   The reference's forward program (mk_bwts_sa.c) needs libdivsufsort, which is
 *   absent from this image, so it is unbuildable here; the suffix sorter below
 *   is our own (SA-IS).  The suffix array is a mathematically unique function of
 *   the text, so any correct sorter is interchangeable at mk_bwts_sa.c:48.
 */
#ifndef BWTS_ORACLE_H
#define BWTS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Suffix array of T[0,n) (no sentinel; a proper prefix sorts first).  Stands in
 * for divsufsort() at mk_bwts_sa.c:48.  Returns 0, or -1 on allocation failure. */
int oracle_suffix_array(const uint8_t *T, int32_t *SA, int64_t n);

/* Forward BWTS following the reference pipeline: SA -> ISA -> Lyndon factors as
 * prefix minima of ISA -> sequential cyclic fix-up -> emission
 * (mk_bwts_sa.c:114-195 and :74-112).  out has n bytes. */
int oracle_forward(const uint8_t *T, int64_t n, uint8_t *out);

/* The same pipeline compiled with 64-bit indices (what oracle_forward uses from 2^31 - 1 bytes on, beyond the
 * reference's int/saidx_t range); callable on any n so that tests can hold it against the 32-bit instance. */
int oracle_forward64(const uint8_t *T, int64_t n, uint8_t *out);

/* Same, also returning per-phase wall seconds in the five MARK_TIME slots of
 * mk_bwts_sa.c:50,124,168,190 (suffix sort, ISA, fix, generate). */
int oracle_forward_timed(const uint8_t *T, int64_t n, uint8_t *out, double phase_s[4]);

/* Forward BWTS straight from the definition (SURVEY.md 8 "Spec"): Duval
 * factorisation, sort every position by rot(p)^omega, emit T[cprev(p)].
 * O(n log n * L); for small n only. */
int oracle_forward_def(const uint8_t *T, int64_t n, uint8_t *out);

/* Inverse BWTS following unbwts.c:31-86 (histogram, exclusive scan, stable LF,
 * cycle walk from the smallest unvisited index, text written backwards). */
int oracle_inverse(const uint8_t *B, int64_t n, uint8_t *out);

/* Lyndon factor starts by Duval's algorithm; writes at most cap starts,
 * returns the number of factors. */
int64_t oracle_lyndon_starts(const uint8_t *T, int64_t n, int64_t *starts, int64_t cap);

/* Synthetic inputs of SURVEY.md 8(d): kind 0 = uniform256, 1 = zipf, 2 = dna; kind 3 = text (the zipf stream with
 * back-references of 16 B .. 64 KiB: the repeat-rich stand-in for enwik8, defined at text_resolve() in the .c file).
 * Bytes [off, off+len) of the stream for the given seed (seekable). */
void oracle_generate(int kind, uint64_t seed, uint64_t off, uint64_t len, uint8_t *dst);

#ifdef __cplusplus
}
#endif
#endif
