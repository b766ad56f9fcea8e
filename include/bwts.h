/*
 * bwts.h -- C-ABI of the MI355X-native bijective-BWT (BWTS) engine (libbwts_hip.so).
 *
 * The reference (NealB/Bijective-BWT) has no FFI: its transform core is inline C
 * on file-scope globals.  These entry points sit exactly on the seams the two
 * reference programs would bind if their cores were swapped for this engine:
 *
 *   bwts_forward()  replaces  divsufsort(T, sa, len) + make_bwts_sa()
 *                             /root/reference/mk_bwts_sa.c:47-52 (core :74-195)
 *   bwts_inverse()  replaces  the histogram / scan / LF / cycle-walk block
 *                             /root/reference/unbwts.c:31-86
 *
 * Plain pointers and sizes only.  Every call returns 0 or a negative BWTS_E_*
 * code and never exits the process (the reference CLIs print and exit(1); the
 * CLIs in bijective-bwt_amd/cli keep that behaviour on top of these codes).
 * A context belongs to one GPU and is used by one host thread at a time;
 * different contexts are independent.
 *
 * There is no CPU fallback: without a usable HIP device bwts_ctx_create() fails
 * with BWTS_E_NODEVICE.
 */
#ifndef BWTS_H
#define BWTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BWTS_OK          0
#define BWTS_E_ARG      -1   /* NULL pointer, n == 0 (the reference fails on empty input: map_file.c:36-40) */
#define BWTS_E_NODEVICE -2   /* no HIP device / device id out of range */
#define BWTS_E_NOMEM    -3   /* device or pinned-host allocation failed */
#define BWTS_E_HIP      -4   /* a HIP runtime call failed; see bwts_last_hip_error() */
#define BWTS_E_RANGE    -5   /* n beyond what the engine indexes: n > 2^36, or one of the documented limits of the blocked 64-bit
                                forward path for n > 2^32 (one 12-bit key prefix larger than a bucket, inputs
                                whose Lyndon factors need the suffix-sort route) */
#define BWTS_E_INTERNAL -6   /* engine invariant violated (bug) */
#define BWTS_E_SINK     -7   /* the caller's output sink returned nonzero */

typedef struct bwts_ctx bwts_ctx;

/* Kernel classes timed with HIP events on the engine's stream. */
enum {
    BWTS_K_HISTOGRAM = 0,   /* byte histogram (forward alphabet / unbwts.c:34-36)        */
    BWTS_K_KEYBUILD,        /* round-0 packed-symbol keys / round-h (rank,rank) keys       */
    BWTS_K_RADIX_HIST,      /* per-tile digit histogram of one LSD pass                   */
    BWTS_K_RADIX_SCAN,      /* exclusive scan of the tile x digit table                   */
    BWTS_K_RADIX_SCATTER,   /* ranked, LDS-staged scatter of one LSD pass (dominant)      */
    BWTS_K_RERANK,          /* group flags + head scan + rank scatter + active compaction */
    BWTS_K_LYNDON,          /* prefix-min scan over suffix ranks -> factor heads          */
    BWTS_K_EMIT,            /* bwts[r] = T[cprev(sa[r])]: patch of tied slots, or full gather (mk_bwts_sa.c:172-188) */
    BWTS_K_LF_BUILD,        /* stable LF map (unbwts.c:50-52)                              */
    BWTS_K_WALK,            /* splitter walk over LF cycles, records segments (unbwts.c:66-86) */
    BWTS_K_LISTRANK,        /* reduced-list ranking of splitter nodes                     */
    BWTS_K_WALK_EMIT,       /* placement of the recorded segments into the text            */
    BWTS_K_OTHER,
    BWTS_K_RADIX_SCATTER_MAIN, /* subset of RADIX_SCATTER: passes 1.. of the n-sized round-0 sort (one kernel variant) */
    BWTS_K_ROUND,           /* group-local round over the tied list, in chunks: gather successor ranks, order every group in LDS,
                               compact in place, apply the new ranks (32 algorithmic bytes per list element and round)          */
    BWTS_K_COUNT
};

#define BWTS_MAX_ROUND_STATS 40

/* Host-side wall-clock costs a context has paid since it was created (what a one-shot CLI run spends outside kernels and copies). */
enum {
    BWTS_H_INIT = 0,        /* bwts_ctx_create: HIP runtime start-up, device, stream, small buffers */
    BWTS_H_MODULE,          /* first kernel launch of the context: the code object is loaded then   */
    BWTS_H_IO_ALLOC,        /* device-side copies of the caller's input / output (hipMalloc)         */
    BWTS_H_STAGING_ALLOC,   /* pinned staging ring + copy workers                                    */
    BWTS_H_ARENA_ALLOC,     /* device arenas (hipMalloc / hipFree when they grow)                    */
    BWTS_H_COUNT
};

typedef struct bwts_kernel_stat {
    double   ms;         /* summed device time of this class in the last call            */
    uint64_t launches;   /* number of launches                                           */
    uint64_t elems;      /* summed elements processed                                    */
    uint64_t alg_bytes;  /* summed ALGORITHMIC bytes, not fetched   */
} bwts_kernel_stat;

typedef struct bwts_timings {
    double   total_ms;            /* device time of the last call, first launch to last   */
    double   h2d_ms, d2h_ms;      /* staging copies (host-buffer entry points only)       */
    uint64_t n;                   /* input length of the last call                        */
    uint64_t factors;             /* forward: Lyndon factors; inverse: LF cycles          */
    uint32_t rounds;              /* forward: sort rounds of the cyclic sort              */
    uint32_t lyndon_rounds;       /* forward: sort rounds spent finding the factors       */
    uint32_t key_symbols;         /* symbols packed into a round-0 key                    */
    uint32_t key_bits;            /* bits of a round-0 key                                */
    uint64_t active_after_round0; /* elements still tied after round 0                    */
    uint64_t unvisited;           /* inverse: elements in cycles without a splitter       */
    uint64_t device_bytes;        /* device memory the context holds after the call (arenas, staging copies of in/out) */
    uint64_t round_active[BWTS_MAX_ROUND_STATS]; /* forward: elements still tied when sort round r+1 starts (r = 0: after round 0) */
    bwts_kernel_stat k[BWTS_K_COUNT];
    double   host_ms[BWTS_H_COUNT]; /* cumulative since bwts_ctx_create (BWTS_H_*): start-up and allocation costs           */
    uint32_t attempts;              /* inverse: times the cycle walk ran (1 on natural inputs, at most 5: see bwts_inverse)  */
    uint32_t reserved_;
} bwts_timings;

int  bwts_ctx_create(bwts_ctx **out, int device_id);
void bwts_ctx_destroy(bwts_ctx *ctx);
/* A context keeps the device memory its calls needed (arena, side blocks, in/out buffers of the host-buffer entry points) for the next
 * call; this hands it back to the device without ending the context.  The next call allocates again.  (No reference counterpart:
 * mk_bwts_sa.c / unbwts.c are one-shot programs that exit.) */
int bwts_ctx_release_memory(bwts_ctx *ctx);

/* Host-buffer entry points: in/out are caller-owned host memory (may be an
 * mmap of a file, unpinned); staged through pinned buffers with hipMemcpyAsync. */
int bwts_forward(bwts_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *out);
int bwts_inverse(bwts_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *out);
/* Cost bounds (n <= 2^32; bwts_timings.device_bytes reports what a context actually holds).  Forward: about 25 n bytes of device
 * memory for inputs whose positions the first sort separates (plus 4 n for the rank array and 16-33 bytes per position that stays
 * tied when it does not: text with long repeats; plus 3 n when the first sort runs on keys wider than 40 bits).  Inverse: about
 * 11 n, plus ~110 bytes per element that lies in a long LF cycle without a splitter (sorted or periodic data); its cycle walk runs
 * once on natural inputs and at most five times on adversarial ones (bwts_timings.attempts). */

/* Same transforms, the result handed to a callback in consecutive pieces (in order, together n bytes) straight from the
 * pinned staging buffers: a CLI passes an fwrite wrapper and never holds an n-byte output buffer (mk_bwts_sa.c:60,
 * unbwts.c:173).  A nonzero return of the sink aborts the call with BWTS_E_SINK. */
typedef int (*bwts_sink_fn)(void *user, const uint8_t *data, uint64_t len);
int bwts_forward_sink(bwts_ctx *ctx, const uint8_t *in, uint64_t n, bwts_sink_fn sink, void *user);
int bwts_inverse_sink(bwts_ctx *ctx, const uint8_t *in, uint64_t n, bwts_sink_fn sink, void *user);

/* Several independent inputs, one after another on this context (what a GPU does in a batched job: BASELINE config 5 puts one such
 * stream of files on every GPU), with the copies overlapped with the transforms: while item k is transformed, item k + 1 is on its
 * way to the device and item k - 1 on its way back.  ins[k] / outs[k]: caller-owned host memory of ns[k] bytes each (unpinned is
 * fine); the bytes are those of count single calls.  outs[k] may lie on ins[k] (in place) or on an earlier item's input; an output that
 * overlaps a LATER item's input is refused with BWTS_E_ARG (that input may not have left the host yet).  Stops at the first error.  Afterwards bwts_last_timings() describes the
 * last item's transform, with d2h_ms = wall time of the whole batch. */
int bwts_forward_batch(bwts_ctx *ctx, int count, const uint8_t *const *ins, const uint64_t *ns, uint8_t *const *outs);
int bwts_inverse_batch(bwts_ctx *ctx, int count, const uint8_t *const *ins, const uint64_t *ns, uint8_t *const *outs);

/* Device-buffer entry points: d_in/d_out are device pointers on the context's
 * GPU (d_out may not alias d_in).  Synchronous: the call returns after the
 * result is complete in d_out. */
int bwts_forward_device(bwts_ctx *ctx, const void *d_in, uint64_t n, void *d_out);
int bwts_inverse_device(bwts_ctx *ctx, const void *d_in, uint64_t n, void *d_out);

/* Statistics of the last forward/inverse call on this context (see bwts_set_timing for the per-kernel times). */
int bwts_last_timings(bwts_ctx *ctx, bwts_timings *t);
const char *bwts_kernel_class_name(int k);
const char *bwts_host_cost_name(int h);

const char *bwts_strerror(int code);
int bwts_last_hip_error(bwts_ctx *ctx);          /* hipError_t of the last BWTS_E_HIP */

/* Per-kernel HIP-event timing (bwts_timings.k[].ms) is off by default: ~400 event records per forward call cost
 * about 1 ms at 1 GiB.  level 0 = off; 1 = only the dominant kernels (BWTS_K_RADIX_SCATTER_MAIN, BWTS_K_ROUND, BWTS_K_WALK:
 * a handful of events); 2 = every class.  Applies to the following calls of this context; the launch / element / byte
 * counters and total_ms are always filled.  A context created while BWTS_TIMINGS=1 is set starts at level 2
 * (the reference's -DSHOW_TIMINGS, mk_bwts_sa.c:13-22). */
int bwts_set_timing(bwts_ctx *ctx, int level);

/* Pinned host memory for callers that want the staging copy out of the way (the CLIs write their output straight
 * from such a buffer): in/out of bwts_forward / bwts_inverse that lie inside a bwts_host_alloc block are transferred
 * by DMA directly.  Optional: any host memory works. */
int bwts_host_alloc(bwts_ctx *ctx, uint64_t bytes, void **h_ptr);
int bwts_host_free(bwts_ctx *ctx, void *h_ptr);

#ifdef __cplusplus
}
#endif
#endif
