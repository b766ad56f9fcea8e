// scan_templ.h -- device-wide scan (reduce / scan-partials / rescan) with functor
// input and output, so flag computation and consumers fuse into the two sweeps.
#pragma once
#include "internal.h"
#include "device_utils.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS   8
#define SCAN_TILE    (SCAN_THREADS * SCAN_ITEMS)

static inline u64 scan_tiles(u64 n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

// Both sweeps touch element i = tile_base + j * SCAN_THREADS + tid (lane-consecutive, coalesced);
// the ops used here are commutative, so the reduce sweep may combine in that order.
template <typename T, typename Op, typename InF>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(u64 n, InF in, Op op, T identity, T *partials)
{
    __shared__ T sm[SCAN_THREADS / 64];
    const u64 base = (u64)blockIdx.x * SCAN_TILE + threadIdx.x;
    T acc = identity;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const u64 i = base + (u64)j * SCAN_THREADS;
        if (i < n) acc = op(acc, in(i));
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

template <typename T, typename Op, typename InF, typename OutF, bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_final_kernel(u64 n, InF in, OutF out, Op op, T identity, const T *partials)
{
    __shared__ T sm[SCAN_THREADS / 64];
    __shared__ T tile[SCAN_TILE + SCAN_TILE / 8];
    const u64 tile_base = (u64)blockIdx.x * SCAN_TILE;
    // striped (coalesced) load -> LDS
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const u32 e = (u32)j * SCAN_THREADS + threadIdx.x;
        const u64 i = tile_base + e;
        tile[SCAN_SLOT(e)] = i < n ? in(i) : identity;
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
        if (i < n) out(i, tile[SCAN_SLOT(e)]);
    }
}

static inline size_t scan_temp_bytes_t(u64 n, size_t elem) { return align_up((size_t)(scan_tiles(n) + 1) * elem, 256); }

// out(i, prefix) is called once per i; in(i) is called twice per i (once per sweep).
template <bool INCLUSIVE, typename T, typename Op, typename InF, typename OutF>
static int device_scan(bwts_ctx *ctx, u64 n, InF in, OutF out, Op op, T identity, void *temp)
{
    if (n == 0) return BWTS_OK;
    const u64 tiles = scan_tiles(n);
    T *partials = (T *)temp;
    scan_reduce_kernel<T, Op, InF><<<dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream>>>(n, in, op, identity, partials);
    scan_partials_kernel<T, Op><<<dim3(1), dim3(1024), 0, ctx->stream>>>(tiles, op, identity, partials);
    scan_final_kernel<T, Op, InF, OutF, INCLUSIVE><<<dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream>>>(n, in, out, op, identity, partials);
    HIPC(hipGetLastError());
    return BWTS_OK;
}
