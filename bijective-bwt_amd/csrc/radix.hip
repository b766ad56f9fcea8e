// radix.hip -- stable 8-bit LSD radix sort of (u64 key, u32 value) pairs.
//
// This replaces the inside of divsufsort() (call site /root/reference/mk_bwts_sa.c:48):
// the engine sorts suffixes / rotations by prefix doubling, and every doubling round is
// an LSD radix sort of packed keys.  One pass = three launches:
//   radix_hist_kernel     per-tile 256-bin digit histogram (LDS, per-wave private bins)
//   column scan           exclusive scan of the [tile][digit] table in digit-major order
//   radix_scatter_kernel  per-wave ballot ranking, LDS-staged tile sort, coalesced scatter
// HBM-bound: algorithmic bytes of the scatter = 2 * (8 + 4) * m per pass.
#include "internal.h"
#include "device_utils.h"
#include "scan_templ.h"

#define RX_THREADS 256
#define RX_WAVES   (RX_THREADS / 64)
#define RX_ITEMS   16
#define RX_TILE    (RX_THREADS * RX_ITEMS)
#define RX_CHUNK   128          // tiles per column-scan chunk

u64 radix_tiles(u64 m) { return (m + RX_TILE - 1) / RX_TILE; }

static inline u64 radix_chunks(u64 tiles) { return (tiles + RX_CHUNK - 1) / RX_CHUNK; }

size_t radix_tile_hist_bytes(u64 m)
{
    const u64 tiles = radix_tiles(m);
    return align_up((size_t)tiles * 256 * sizeof(u32), 256) +
           align_up((size_t)radix_chunks(tiles) * 256 * sizeof(u32), 256);
}

// ------------------------------------------------------------------------------------
// generic in-place exclusive sum (used for the chunk table and elsewhere)
// ------------------------------------------------------------------------------------
struct LoadU32  { const u32 *p; __device__ __forceinline__ u32 operator()(u64 i) const { return p[i]; } };
struct StoreU32 { u32 *p; __device__ __forceinline__ void operator()(u64 i, u32 v) const { p[i] = v; } };

size_t scan_temp_bytes(u64 n) { return scan_temp_bytes_t(n, sizeof(u64)); }

int exclusive_sum_u32(bwts_ctx *ctx, u32 *data, u64 n, void *temp)
{
    LoadU32 in{data};
    StoreU32 out{data};
    return device_scan<false, u32>(ctx, n, in, out, OpAdd(), (u32)0, temp);
}

// ------------------------------------------------------------------------------------
// pass kernels
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(RX_THREADS) void radix_hist_kernel(const u64 *__restrict__ keys, u64 m, int shift,
                                                                 u32 *__restrict__ tile_hist)
{
    __shared__ u32 bins[RX_WAVES][256];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < RX_WAVES * 256; i += RX_THREADS) ((u32 *)bins)[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * RX_TILE;
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const u64 i = base + (u64)j * RX_THREADS + tid;
        if (i < m) {
            const u32 d = (u32)(keys[i] >> shift) & 255u;
            atomicAdd(&bins[w][d], 1u);
        }
    }
    __syncthreads();
    u32 s = 0;
#pragma unroll
    for (int ww = 0; ww < RX_WAVES; ww++) s += bins[ww][tid];
    tile_hist[(u64)blockIdx.x * 256 + tid] = s;
}

// column sums of a chunk of tiles: chunk_sum[d * chunks + c]
__global__ __launch_bounds__(256) void radix_chunk_sum_kernel(const u32 *__restrict__ tile_hist, u64 tiles, u64 chunks,
                                                              u32 *__restrict__ chunk_sum)
{
    const u64 c = blockIdx.x;
    const u64 t0 = c * RX_CHUNK;
    const u64 t1 = t0 + RX_CHUNK < tiles ? t0 + RX_CHUNK : tiles;
    u32 s = 0;
    for (u64 t = t0; t < t1; t++) s += tile_hist[t * 256 + threadIdx.x];
    chunk_sum[(u64)threadIdx.x * chunks + c] = s;
}

// turns per-tile counts into global exclusive offsets, in place
__global__ __launch_bounds__(256) void radix_chunk_apply_kernel(u32 *__restrict__ tile_hist, u64 tiles, u64 chunks,
                                                                const u32 *__restrict__ chunk_off)
{
    const u64 c = blockIdx.x;
    const u64 t0 = c * RX_CHUNK;
    const u64 t1 = t0 + RX_CHUNK < tiles ? t0 + RX_CHUNK : tiles;
    u32 run = chunk_off[(u64)threadIdx.x * chunks + c];
    for (u64 t = t0; t < t1; t++) {
        const u32 v = tile_hist[t * 256 + threadIdx.x];
        tile_hist[t * 256 + threadIdx.x] = run;
        run += v;
    }
}

__global__ __launch_bounds__(RX_THREADS) void radix_scatter_kernel(const u64 *__restrict__ kin, const u32 *__restrict__ vin,
                                                                    u64 *__restrict__ kout, u32 *__restrict__ vout,
                                                                    const u32 *__restrict__ tile_off, u64 m, int shift)
{
    __shared__ u64 skeys[RX_TILE];
    __shared__ u32 svals[RX_TILE];
    __shared__ u32 whist[RX_WAVES][256];
    __shared__ u32 dbase[256];
    __shared__ u32 gbase[256];
    __shared__ u32 scan_sm[RX_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u64 tile_base = (u64)blockIdx.x * RX_TILE;
    const u64 wave_base = tile_base + (u64)w * (64 * RX_ITEMS);
    const u64 remain = m - tile_base;
    const u32 tile_count = remain < RX_TILE ? (u32)remain : (u32)RX_TILE;

    for (int i = tid; i < RX_WAVES * 256; i += RX_THREADS) ((u32 *)whist)[i] = 0;

    u64 key[RX_ITEMS];
    u32 val[RX_ITEMS];
    u32 rnk[RX_ITEMS];
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const u64 i = wave_base + (u64)j * 64 + lane;
        const bool valid = i < m;
        key[j] = valid ? kin[i] : ~0ull;
        val[j] = valid ? vin[i] : 0u;
    }
    __syncthreads();

    // stable rank of every element among equal digits of its wave (element order = (j, lane))
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const bool valid = wave_base + (u64)j * 64 + lane < m;
        const u32 d = (u32)(key[j] >> shift) & 255u;
        const u64 peers = match_digit8(d, valid);
        const u32 before = (u32)__popcll(peers & lanemask_lt());
        const u32 cnt = (u32)__popcll(peers);
        const u32 prev = whist[w][d];
        rnk[j] = prev + before;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid && before == 0) whist[w][d] = prev + cnt;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();

    // per digit: exclusive offsets across waves, then across digits
    {
        u32 run = 0;
#pragma unroll
        for (int ww = 0; ww < RX_WAVES; ww++) {
            const u32 c = whist[ww][tid];
            whist[ww][tid] = run;
            run += c;
        }
        u32 total;
        const u32 exc = block_scan_exclusive<u32, OpAdd, RX_WAVES>(run, OpAdd(), 0u, scan_sm, &total);
        dbase[tid] = exc;
        gbase[tid] = tile_off[(u64)blockIdx.x * 256 + tid] - exc;
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const bool valid = wave_base + (u64)j * 64 + lane < m;
        if (valid) {
            const u32 d = (u32)(key[j] >> shift) & 255u;
            const u32 pos = dbase[d] + whist[w][d] + rnk[j];
            skeys[pos] = key[j];
            svals[pos] = val[j];
        }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const u32 s = (u32)j * RX_THREADS + tid;
        if (s < tile_count) {
            const u64 k = skeys[s];
            const u32 d = (u32)(k >> shift) & 255u;
            const u32 dst = gbase[d] + s;
            kout[dst] = k;
            vout[dst] = svals[s];
        }
    }
}

// ------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------
// [tile][digit] counts -> global exclusive offsets in digit-major order, in place.
// tile_hist must be followed by the chunk table (radix_tile_hist_bytes()).
int radix_column_scan(bwts_ctx *ctx, u32 *tile_hist, u64 tiles, void *scan_temp)
{
    const u64 chunks = radix_chunks(tiles);
    u32 *chunk_sum = (u32 *)((char *)tile_hist + align_up((size_t)tiles * 256 * sizeof(u32), 256));
    radix_chunk_sum_kernel<<<dim3((unsigned)chunks), dim3(256), 0, ctx->stream>>>(tile_hist, tiles, chunks, chunk_sum);
    BWTS_TRY(exclusive_sum_u32(ctx, chunk_sum, chunks * 256, scan_temp));
    radix_chunk_apply_kernel<<<dim3((unsigned)chunks), dim3(256), 0, ctx->stream>>>(tile_hist, tiles, chunks, chunk_sum);
    HIPC(hipGetLastError());
    return BWTS_OK;
}

int radix_sort_pairs(bwts_ctx *ctx, const SortPlan &plan, u64 m, int key_bits, int *result_buf)
{
    int cur = 0;
    if (m == 0) { *result_buf = 0; return BWTS_OK; }
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 64) key_bits = 64;
    const int passes = (key_bits + 7) / 8;
    const u64 tiles = radix_tiles(m);
    u32 *tile_hist = plan.tile_hist;

    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        {
            SpanGuard g(ctx, BWTS_K_RADIX_HIST, m, 8 * m);
            radix_hist_kernel<<<dim3((unsigned)tiles), dim3(RX_THREADS), 0, ctx->stream>>>(plan.keys[cur], m, shift, tile_hist);
        }
        {
            SpanGuard g(ctx, BWTS_K_RADIX_SCAN, tiles * 256, tiles * 256 * 12);
            BWTS_TRY(radix_column_scan(ctx, tile_hist, tiles, plan.scan_temp));
        }
        {
            SpanGuard g(ctx, BWTS_K_RADIX_SCATTER, m, 24 * m);
            radix_scatter_kernel<<<dim3((unsigned)tiles), dim3(RX_THREADS), 0, ctx->stream>>>(
                plan.keys[cur], plan.vals[cur], plan.keys[cur ^ 1], plan.vals[cur ^ 1], tile_hist, m, shift);
        }
        HIPC(hipGetLastError());
        cur ^= 1;
    }
    *result_buf = cur;
    return BWTS_OK;
}
