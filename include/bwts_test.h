/*
 * bwts_test.h -- harness and unit-test entry points of libbwts_hip.so.
 *
 * Not part of the drop-in surface (include/bwts.h is): these exist for tests/, bench.py and tools/ --
 * synthetic inputs generated in device memory, raw device buffers without a tensor library, and hooks that
 * run single stages of the engine.  The reference (NealB/Bijective-BWT) has no counterpart.
 */
#ifndef BWTS_TEST_H
#define BWTS_TEST_H

#include "bwts.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Harness utilities (bench / tests): synthetic inputs of SURVEY.md 8(d) written
 * straight into device memory (kind 0 uniform256, 1 zipf, 2 dna, 3 text: zipf stream with back-references), device
 * buffers without a tensor library, and a 64-bit FNV-style checksum. */
int bwts_generate_device(bwts_ctx *ctx, int kind, uint64_t seed, uint64_t n, void *d_out);
int bwts_device_alloc(bwts_ctx *ctx, uint64_t bytes, void **d_ptr);
int bwts_device_free(bwts_ctx *ctx, void *d_ptr);
int bwts_copy_to_device(bwts_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int bwts_copy_to_host(bwts_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
int bwts_device_equal(bwts_ctx *ctx, const void *d_a, const void *d_b, uint64_t bytes, int *equal);

/* Unit-test hooks for single kernels (stable LSD radix sort of (u64 key, u32
 * value) pairs on the low key_bits bits; suffix array via the non-cyclic sort). */
int bwts_debug_sort_pairs(bwts_ctx *ctx, uint64_t *h_keys, uint32_t *h_vals, uint64_t m, int key_bits);
int bwts_debug_suffix_array(bwts_ctx *ctx, const uint8_t *in, uint64_t n, uint32_t *h_sa);
int bwts_debug_lyndon(bwts_ctx *ctx, const uint8_t *in, uint64_t n, uint64_t *h_starts, uint64_t cap, uint64_t *count);

#ifdef __cplusplus
}
#endif
#endif
