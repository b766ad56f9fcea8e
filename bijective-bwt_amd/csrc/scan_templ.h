// scan_templ.h -- device-wide scan (reduce / scan-partials / rescan) with functor
// input and output, so flag computation and consumers fuse into the two sweeps.
#pragma once
#include "internal.h"
#include "device_utils.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS   8
#define SCAN_TILE    (SCAN_THREADS * SCAN_ITEMS)
#define SCAN_SINGLE_BLOCK_MAX 8192ull     // partial counts up to this are scanned by one workgroup

static inline u64 scan_tiles(u64 n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

// functors either take (i) / (i, value) or, to pass a note from load to store, (i, u32*) / (i, value, u32)
template <int N> struct scan_rank : scan_rank<N - 1> {};
template <> struct scan_rank<0> {};
template <typename F> __device__ __forceinline__ auto scan_call_in_(const F &f, u64 i, u32 *note, scan_rank<1>) -> decltype(f(i, note)) { return f(i, note); }
template <typename F> __device__ __forceinline__ auto scan_call_in_(const F &f, u64 i, u32 *, scan_rank<0>) -> decltype(f(i)) { return f(i); }
template <typename F, typename T> __device__ __forceinline__ auto scan_call_out_(const F &f, u64 i, T v, u32 note, scan_rank<1>) -> decltype(f(i, v, note)) { return f(i, v, note); }
template <typename F, typename T> __device__ __forceinline__ auto scan_call_out_(const F &f, u64 i, T v, u32, scan_rank<0>) -> decltype(f(i, v)) { return f(i, v); }
#define scan_call_in(f, i, note) scan_call_in_(f, i, note, scan_rank<1>())

// A functor may split its work into load(i) -> Raw (global loads only) and make(i, raw, note) -> T, so that the
// kernels can issue all of a thread's loads back to back before any of them is consumed.
template <typename F> struct scan_raw_of {
    template <typename G> static auto probe(int) -> decltype(((const G *)nullptr)->load((u64)0));
    template <typename G> static char probe(...);
    typedef decltype(probe<F>(0)) type;       // char = no load()
};
template <typename F, typename T>
__device__ __forceinline__ auto scan_load_(const F &f, u64 i, scan_rank<1>) -> decltype(f.load(i)) { return f.load(i); }
template <typename F, typename T>
__device__ __forceinline__ char scan_load_(const F &, u64, scan_rank<0>) { return 0; }
template <typename F, typename R, typename T>
__device__ __forceinline__ auto scan_make_(const F &f, u64 i, const R &raw, u32 *note, scan_rank<1>) -> decltype(f.make(i, raw, note)) { return f.make(i, raw, note); }
template <typename F, typename R, typename T>
__device__ __forceinline__ T scan_make_(const F &f, u64 i, const R &, u32 *note, scan_rank<0>) { return scan_call_in(f, i, note); }
#define scan_call_out(f, i, v, note) scan_call_out_(f, i, v, note, scan_rank<1>())

// Both sweeps touch element i = tile_base + j * SCAN_THREADS + tid (lane-consecutive, coalesced);
// the ops used here are commutative, so the reduce sweep may combine in that order.
template <typename T, typename Op, typename InF>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(u64 n, InF in, Op op, T identity, T *partials)
{
    __shared__ T sm[SCAN_THREADS / 64];
    const u64 base = (u64)blockIdx.x * SCAN_TILE + threadIdx.x;
    // The input functor is evaluated for EVERY item, at an index clamped into the range, and the value of an item past the end is
    // dropped: under `if (i < n)` each item's loads sat behind a branch with a wait for the data right after them -- one load in flight
    // per wave, in every scan of the library.  (Input functors only read.)
    T acc = identity;
    typename scan_raw_of<InF>::type raw[SCAN_ITEMS];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const u64 i = base + (u64)j * SCAN_THREADS;
        raw[j] = scan_load_<InF, T>(in, i < n ? i : n - 1, scan_rank<1>());
    }
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const u64 i = base + (u64)j * SCAN_THREADS;
        u32 note;
        const T v = scan_make_<InF, typename scan_raw_of<InF>::type, T>(in, i < n ? i : n - 1, raw[j], &note, scan_rank<1>());
        acc = i < n ? op(acc, v) : acc;
    }
    T inc = wave_scan_inclusive(acc, op);
    if (lane_id() == 63) sm[wave_id()] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = identity;
        for (int w = 0; w < SCAN_THREADS / 64; w++) t = op(t, sm[w]);
        partials[blockIdx.x] = t;
    }
}

// one workgroup walks all partials with a running carry; exclusive, in place
template <typename T, typename Op>
__global__ __launch_bounds__(1024) void scan_partials_kernel(u64 count, Op op, T identity, T *partials)
{
    __shared__ T sm[16];
    T carry = identity;
    for (u64 base = 0; base < count; base += 1024) {
        const u64 i = base + threadIdx.x;
        T v = i < count ? partials[i] : identity;
        T tot;
        T exc = block_scan_exclusive<T, Op, 16>(v, op, identity, sm, &tot);
        if (i < count) partials[i] = op(carry, exc);
        carry = op(carry, tot);
    }
}

// LDS slot of tile element e: one pad slot per 8 keeps the blocked (8 per thread) accesses
// spread over the banks
#define SCAN_SLOT(e) ((e) + ((e) >> 3))

struct ScanAllTiles { template <typename T> __device__ __forceinline__ bool operator()(u64, T) const { return true; } };

// filter(tile, carry_in): a false answer skips the tile (no loads, no calls of in/out)
template <typename T, typename Op, typename InF, typename OutF, bool INCLUSIVE, typename Filter>
__global__ __launch_bounds__(SCAN_THREADS) void scan_final_kernel(u64 n, InF in, OutF out, Op op, T identity, const T *partials, Filter filter)
{
    __shared__ T sm[SCAN_THREADS / 64];
    __shared__ T tile[SCAN_TILE + SCAN_TILE / 8];
    if (!filter((u64)blockIdx.x, partials[blockIdx.x])) return;
    const u64 tile_base = (u64)blockIdx.x * SCAN_TILE;
    // striped (coalesced) load -> LDS; a functor may hand a 32-bit note from its load to its store
    // (both see the same element in the same lane), e.g. flags that cost a global load to recompute
    u32 note[SCAN_ITEMS];
    {
        typename scan_raw_of<InF>::type raw[SCAN_ITEMS];
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {
            const u64 i = tile_base + (u32)j * SCAN_THREADS + threadIdx.x;
            raw[j] = scan_load_<InF, T>(in, i < n ? i : n - 1, scan_rank<1>());        // (clamped, unconditional: see scan_reduce_kernel)
        }
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {
            const u32 e = (u32)j * SCAN_THREADS + threadIdx.x;
            const u64 i = tile_base + e;
            note[j] = 0;
            const T v = scan_make_<InF, typename scan_raw_of<InF>::type, T>(in, i < n ? i : n - 1, raw[j], &note[j], scan_rank<1>());
            tile[SCAN_SLOT(e)] = i < n ? v : identity;
        }
    }
    __syncthreads();
    // blocked scan: thread t owns elements [8t, 8t+8)
    T v[SCAN_ITEMS];
    T acc = identity;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        v[j] = tile[SCAN_SLOT((u32)threadIdx.x * SCAN_ITEMS + j)];
        acc = op(acc, v[j]);
    }
    T tot;
    T run = block_scan_exclusive<T, Op, SCAN_THREADS / 64>(acc, op, identity, sm, &tot);
    run = op(partials[blockIdx.x], run);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const u32 slot = SCAN_SLOT((u32)threadIdx.x * SCAN_ITEMS + j);
        if (INCLUSIVE) {
            run = op(run, v[j]);
            tile[slot] = run;
        } else {
            tile[slot] = run;
            run = op(run, v[j]);
        }
    }
    __syncthreads();
    // striped (coalesced) consumer
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const u32 e = (u32)j * SCAN_THREADS + threadIdx.x;
        const u64 i = tile_base + e;
        if (i < n) scan_call_out(out, i, tile[SCAN_SLOT(e)], note[j]);
    }
}

// partials of every level: tiles + tiles/2048 + ... (each level padded)
static inline size_t scan_temp_bytes_t(u64 n, size_t elem)
{
    size_t total = 0;
    for (u64 t = scan_tiles(n);; t = scan_tiles(t)) {
        total += align_up((size_t)(t + 1) * elem, 256);
        if (t <= SCAN_SINGLE_BLOCK_MAX) break;
    }
    return total;
}

template <typename T> struct ScanLoadArr  { const T *p; __device__ __forceinline__ T operator()(u64 i) const { return p[i]; } };
template <typename T> struct ScanStoreArr { T *p; __device__ __forceinline__ void operator()(u64 i, T v) const { p[i] = v; } };

// exclusive scan of the per-tile reductions already sitting in `temp`
template <typename T, typename Op>
static int device_scan_partials(bwts_ctx *ctx, u64 tiles, Op op, T identity, void *temp);

// final sweep only: `temp` holds the scanned per-tile carries
template <bool INCLUSIVE, typename T, typename Op, typename InF, typename OutF, typename Filter>
static int device_scan_final(bwts_ctx *ctx, u64 n, InF in, OutF out, Op op, T identity, void *temp, Filter filter)
{
    if (n == 0) return BWTS_OK;
    scan_final_kernel<T, Op, InF, OutF, INCLUSIVE, Filter><<<dim3((unsigned)scan_tiles(n)), dim3(SCAN_THREADS), 0, ctx->stream>>>(
        n, in, out, op, identity, (const T *)temp, filter);
    HIPC(hipGetLastError());
    return BWTS_OK;
}

// out(i, prefix) is called once per i; in(i) is called twice per i (once per sweep).
template <bool INCLUSIVE, typename T, typename Op, typename InF, typename OutF>
static int device_scan(bwts_ctx *ctx, u64 n, InF in, OutF out, Op op, T identity, void *temp)
{
    if (n == 0) return BWTS_OK;
    const u64 tiles = scan_tiles(n);
    T *partials = (T *)temp;
    scan_reduce_kernel<T, Op, InF><<<dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream>>>(n, in, op, identity, partials);
    BWTS_TRY((device_scan_partials<T, Op>(ctx, tiles, op, identity, temp)));
    return device_scan_final<INCLUSIVE, T>(ctx, n, in, out, op, identity, temp, ScanAllTiles());
}

template <typename T, typename Op>
static int device_scan_partials(bwts_ctx *ctx, u64 tiles, Op op, T identity, void *temp)
{
    T *partials = (T *)temp;
    if (tiles <= SCAN_SINGLE_BLOCK_MAX) {
        scan_partials_kernel<T, Op><<<dim3(1), dim3(1024), 0, ctx->stream>>>(tiles, op, identity, partials);
        HIPC(hipGetLastError());
        return BWTS_OK;
    }
    // many partials: scan them with the same three-kernel scheme one level up (exclusive, in place)
    void *next = (char *)temp + align_up((size_t)(tiles + 1) * sizeof(T), 256);
    ScanLoadArr<T> pin{partials};
    ScanStoreArr<T> pout{partials};
    return device_scan<false, T>(ctx, tiles, pin, pout, op, identity, next);
}
