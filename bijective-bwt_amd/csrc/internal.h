// internal.h -- host-side plumbing shared by the engine's translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <atomic>

#include "../../include/bwts.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t  u8;

struct TimedSpan {
    int cls;
    hipEvent_t a, b;
};

// Host-side copy workers for the host-buffer entry points: a chunk is cut into page-aligned parts and every worker (and the
// caller) memcpy's one.  One thread moves ~10 GB/s into or out of pinned staging -- and pays every first-touch page fault
// of a fresh output buffer alone -- which is below what the PCIe link carries.
struct CopyPool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    u64 generation = 0;
    int pending = 0;
    bool stop = false;
    char *dst = nullptr;
    const char *src = nullptr;
    size_t len = 0;
    int parts = 1;
    bool touching = false;  // the current job pre-faults the pages of dst (madvise, contents untouched) instead of copying
    std::atomic<bool> touch_failed{false};
    void start(int threads);
    void copy(void *dst, const void *src, size_t len);      // returns when all parts are done
    // The workers alone fault in the pages of a fresh buffer (MADV_POPULATE_WRITE: mapped writable, contents as they were) while
    // the caller goes on: the first-touch faults of a caller's output buffer are paid while the GPU transforms.  wait() before the
    // next job.
    void touch_async(void *dst, size_t len);
    void wait();
    void shutdown();
    void part(int i);
};

#define STAGE_SLOTS 4
#define BWTS_AUX_SLOTS 5

#define SM_RX_SYNC 4090          // d_small word holding the two 32-bit counters of radix_column_scan_fused_kernel (zero between launches)

struct Stager {
    hipStream_t stream;             // the queue its copies (and copy kernels) are issued on
    bool own_stream;
    char *pinned[STAGE_SLOTS];
    hipEvent_t slot_ev[STAGE_SLOTS], slot_ev2[STAGE_SLOTS];
    hipStream_t copy_stream;        // second queue: part of a device-to-host chunk goes through the DMA engine while a kernel moves the rest
    CopyPool *pool;
};

struct bwts_ctx {
    int device;
    hipStream_t stream;
    size_t call_block_bytes = 0;        // device memory taken for one call only (rare paths), largest of the last call
    int last_hip;

    // device arena: one allocation, bump-allocated per call, grown between calls
    char  *arena;
    size_t arena_cap, arena_off;
    // side arenas sized on demand.  0, 1: forward: tied-set buffers; inverse: unreached-element lists, cycle sort.  2: factor list of
    // the general Lyndon path.  3: previous-symbol array + carried-byte buffers (rounds-0 sorts on wide keys).  4: dense rank array.
    // (What the headline path does not touch is not allocated: the driver clears device memory it hands out, ~27 ms per GiB.)
    char  *aux[BWTS_AUX_SLOTS];
    size_t aux_cap[BWTS_AUX_SLOTS];
    size_t unv_hint;       // inverse: unreached elements seen by the previous call (sizes the first collection pass)
    // tied list of the forward transform beyond 2^32 positions (wide_path.h): blocks of 2^tied_blk_lg (position, head) pairs, taken as the
    // list grows and kept for the next call
    std::vector<char *> tied_blk;
    int tied_blk_lg = 0;
    // BWTS_GUARD=1 (a test switch): every block the context takes from the device gets `guard` bytes of a fixed pattern in front and
    // behind, checked after every transform: a kernel that writes outside its buffers is named, with the block and the offset
    size_t guard = 0;
    struct GuardBlock { char *user; size_t bytes; const char *name; };
    std::vector<GuardBlock> guard_blocks;
    std::vector<GuardBlock> guard_freed;         // ... and what the context has given up stays mapped, filled with a second pattern: a write through a stale pointer shows

    // small pinned host block for read-backs, and a device mirror
    u64 *h_small;          // 4096 u64
    u64 *d_small;          // 4096 u64

    // host-buffer entry points: staging (a ring of pinned slots with one event each, copy workers, a queue of its own), device-side
    // in/out buffers that stay with the context, and the pinned blocks handed out by bwts_host_alloc.  stg[0] serves the single
    // calls on the context's own stream; the batch entry points move the neighbouring items' data on stg[1] (in) and stg[2] (out)
    // while the current item is transformed, between two pairs of device buffers (d_io[0..1] in, d_io[2..3] out).
    Stager stg[3];
    u8    *d_io[4];
    size_t d_io_cap[4];
    std::vector<std::pair<char *, size_t>> host_blocks;

    // event pool + spans of the current call
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used;
    std::vector<TimedSpan> spans;
    hipEvent_t ev_begin, ev_end;
    int timing;            // HIP-event spans: 0 none (counters only), 1 dominant kernels, 2 all (bwts_set_timing / BWTS_TIMINGS=1)

    // kernels that were granted more than 64 KB of dynamic LDS on this context's device (the attribute is per device)
    std::vector<const void *> lds_granted;

    // environment knobs, read ONCE when the context is made (bwts_knob()): diagnostics and staging tuning always, the switches that
    // select alternate code paths -- what the test suite drives -- only when BWTS_TEST_KNOBS=1 is set
    std::vector<std::pair<std::string, std::string>> knobs;
    int fused_scan_cap;    // workgroups of radix_column_scan_fused_kernel that are certain to be resident together (0 = not yet asked)
    int chain_cap;         // ... of the chained packed pass kernel (radix.hip, CHAINED passes)
    int rx_config;         // radix tile shape (BWTS_RX_CONFIG, a tuning knob; 0 = the product shape)

    bwts_timings tm;
    double host_ms[BWTS_H_COUNT];   // cumulative host-side costs (BWTS_H_*)
    bool launched;                  // a kernel of this context has run (the code object is loaded)
};

// BWTS_STAGE_TRACE=1 (diagnosis): the stream is drained after every stage of a transform and the stage is named on stderr, so that
// a GPU fault -- reported asynchronously -- is known to come from the stage after the last one named
void bwts_stage_mark(bwts_ctx *ctx, const char *name);
#define STAGE(name) bwts_stage_mark(ctx, name)

#define HIPC(call)                                                        \
    do {                                                                  \
        hipError_t e__ = (call);                                          \
        if (e__ != hipSuccess) { ctx->last_hip = (int)e__; return BWTS_E_HIP; } \
    } while (0)

// BWTS_TRACE_ERRORS=1: every failing call on the way up prints its place (a debugging aid; costs nothing until something fails)
void bwts_trace_error(const char *file, int line, int rc);
#define BWTS_TRY(call)                   \
    do {                                 \
        int rc__ = (call);               \
        if (rc__ != BWTS_OK) { bwts_trace_error(__FILE__, __LINE__, rc__); return rc__; } \
    } while (0)

const char *bwts_knob(const bwts_ctx *ctx, const char *name);    // value of an environment knob as the context saw it, or null

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
double wall_ms(void);

// ---- arena -------------------------------------------------------------------
int  arena_reserve(bwts_ctx *ctx, size_t bytes);        // (re)allocates when too small; resets
int  arena_release(bwts_ctx *ctx);                      // drains the context's stream, then gives the arena up (calling thread only)
void arena_install(bwts_ctx *ctx, void *block, size_t bytes, double alloc_ms);
void arena_reset(bwts_ctx *ctx);
void *arena_alloc(bwts_ctx *ctx, size_t bytes);         // NULL when exhausted
template <typename T> static inline T *arena_array(bwts_ctx *ctx, u64 count)
{
    return (T *)arena_alloc(ctx, (size_t)count * sizeof(T));
}

// opt-in for > 64 KB of dynamic LDS, once per kernel and context (a context is bound to one device)
int ensure_dyn_lds(bwts_ctx *ctx, const void *kernel, size_t bytes);

// ---- timing ------------------------------------------------------------------
int  span_begin(bwts_ctx *ctx, int cls, u64 elems, u64 alg_bytes);   // returns a span handle (spans may nest)
void span_end(bwts_ctx *ctx, int span);
void spans_reset(bwts_ctx *ctx);
int  spans_resolve(bwts_ctx *ctx);

struct SpanGuard {
    bwts_ctx *c;
    int h;
    SpanGuard(bwts_ctx *ctx, int cls, u64 elems, u64 alg_bytes) : c(ctx), h(span_begin(ctx, cls, elems, alg_bytes)) {}
    ~SpanGuard() { span_end(c, h); }
};

// ---- read-back of a few u64 words (synchronises the stream) --------------------
int read_small(bwts_ctx *ctx, int first, int count);    // d_small[first..) -> h_small[first..)

// ---- device-wide primitives (scan.hip) ------------------------------------------
size_t scan_temp_bytes(u64 n);
int exclusive_sum_u32(bwts_ctx *ctx, u32 *data, u64 n, void *temp);   // in place, wraps mod 2^32

// ---- LSD radix sort of (u64 key, u32 value) pairs (radix.hip) -------------------
struct SortPlan {
    u64  *keys[2];
    u32  *vals[2];
    u32  *tile_hist;      // 256 * tiles(m)
    void *scan_temp;
    // optional third stream: one byte per element travels with the pair (all three may be null)
    const u8 *sym_src = nullptr;     // byte of element i before the first pass
    u8 *sym_buf[2] = {nullptr, nullptr};
    u8 *sym_final = nullptr;         // where the last pass leaves the bytes
    bool vals_identity = false;      // vals[0] is not read: the first pass uses value = element index
    // keys[0] holds the keys split for the packed sort (radix_packed_applicable()): u32 low words at keys[0], and at
    // (u8 *)keys[0] + align_up(4 m, 256) the c stream: u16 (key bits 32..39 | carried byte << 8) for keys of more than 32
    // bits, else u8 (the carried byte).  sym_src is not read; needs sym_final and vals_identity.
    bool keys_split = false;
};
bool radix_packed_applicable(const bwts_ctx *ctx, u64 m, int key_bits);   // will radix_sort_pairs run its packed-stream passes for such a sort?
bool radix_supports_sym(const bwts_ctx *ctx);       // the byte stream is compiled for the default tile shape only
u64    radix_tiles(const bwts_ctx *ctx, u64 m);
size_t radix_tile_hist_bytes(u64 m);
// Sorts on key bits [0, key_bits); returns in *result_buf which of keys[]/vals[] holds the output.
int radix_sort_pairs(bwts_ctx *ctx, const SortPlan &plan, u64 m, int key_bits, int *result_buf);
int radix_column_scan(bwts_ctx *ctx, u32 *tile_hist, u64 tiles, void *scan_temp);
int radix_sort_keys(bwts_ctx *ctx, u64 *keys[2], u32 *tile_hist, void *scan_temp, u64 m, int lo_bit, int bits, int *result_buf);

// ---- forward / inverse drivers ----------------------------------------------------
int forward_device_impl(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out);
int inverse_device_impl(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out);
size_t forward_arena_bytes(u64 n);
size_t inverse_arena_bytes(u64 n);

// non-cyclic suffix sort: leaves the suffix array in *d_sa (arena memory) and ranks in *d_rank
int suffix_sort_device(bwts_ctx *ctx, const u8 *d_T, u64 n, u32 **d_sa, u32 **d_rank, u32 *rounds);
int lyndon_factors_device(bwts_ctx *ctx, const u8 *d_T, u64 n, u32 **d_fstart, u64 *k, u32 *rounds);
int byte_histogram_device(bwts_ctx *ctx, const u8 *d_T, u64 n, u64 *d_hist256);
int constant_input_probe(bwts_ctx *ctx, const u8 *d_in, u64 n, bool *constant);   // one byte value repeated?
int aux_reserve_slot(bwts_ctx *ctx, int slot, size_t bytes, char **base);   // side arenas, sized on demand
static inline int aux_reserve(bwts_ctx *ctx, size_t bytes, char **base) { return aux_reserve_slot(ctx, 0, bytes, base); }

// ---- generators / utilities (gen.hip) ------------------------------------------------
int generate_device_impl(bwts_ctx *ctx, int kind, u64 seed, u64 n, u8 *d_out);
int device_equal_impl(bwts_ctx *ctx, const u8 *a, const u8 *b, u64 bytes, int *equal);
