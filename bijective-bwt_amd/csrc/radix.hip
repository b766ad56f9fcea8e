// radix.hip -- stable 8-bit LSD radix sort of (u64 key, u32 value) pairs, with an optional byte stream riding along.
//
// This replaces the inside of divsufsort() (call site /root/reference/mk_bwts_sa.c:48):
// the engine sorts suffixes / rotations by prefix doubling, and every doubling round is
// an LSD radix sort of packed keys.  One pass = three launches:
//   histogram kernel      per-tile 256-bin digit histogram (LDS, per-wave private bins)
//   column scan           exclusive scan of the [tile][digit] table in digit-major order
//   scatter kernel        per-wave ballot ranking, LDS-staged tile sort, coalesced scatter
// Two families of pass kernels:
//   radix_hist_kernel / radix_scatter2_kernel                wide streams: u64 key, u32 value (+ u8 byte); 2*(8+4[+1]) bytes
//                                                            per element and pass.  Later rounds, keys wider than 40 bits.
//   radix_hist_packed_kernel / radix_scatter_packed_kernel   round 0 of the forward transform with keys of <= 40 bits:
//                                                            packed streams, 2*(4+4+2) bytes (see "packed streams" below).
#include "internal.h"
#include "device_utils.h"
#include "scan_templ.h"

#include <stdlib.h>

#define RX_CHUNK   128          // tiles per column-scan chunk
#define RX_XCDS    8

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), and each XCD has its own
// non-coherent L2.  Tile t's run for digit d is followed in memory by tile t+1's run, so the two share a
// cache line at the seam: give every XCD a CONTIGUOUS range of tiles, walked in dispatch order, so that
// both halves of a seam line are written through the same L2 close in time and leave it as one full line.
// Placement only changes speed, never results.
__device__ __forceinline__ u64 rx_tile_of_block(u64 tiles)
{
    const u64 per = (tiles + RX_XCDS - 1) / RX_XCDS;
    return (u64)(blockIdx.x % RX_XCDS) * per + (u64)(blockIdx.x / RX_XCDS);
}
static inline u64 rx_grid(u64 tiles) { return (tiles + RX_XCDS - 1) / RX_XCDS * RX_XCDS; }

// tile shapes compiled in; BWTS_RX_CONFIG picks one at first use (tuning knob).  Shape 0 is the product shape:
// 512 threads x 16 items = 8192 elements, two tiles resident per CU; only it carries the byte stream.
struct RxConfig { int threads, items; };
static const RxConfig kRxConfigs[] = {{512, 16}, {512, 12}, {256, 16}, {1024, 8}};
#define RX_NCONFIGS ((int)(sizeof(kRxConfigs) / sizeof(kRxConfigs[0])))
#define RX_DEFAULT_CONFIG 0

static int rx_config_index(const bwts_ctx *ctx) { return ctx->rx_config; }      // parsed from BWTS_RX_CONFIG when the context was made (api.hip)
static u64 rx_tile(const bwts_ctx *ctx) { const RxConfig &c = kRxConfigs[rx_config_index(ctx)]; return (u64)c.threads * c.items; }

int radix_config_count(void) { return RX_NCONFIGS; }
u64 radix_tiles(const bwts_ctx *ctx, u64 m) { return (m + rx_tile(ctx) - 1) / rx_tile(ctx); }

static inline u64 radix_chunks(u64 tiles) { return (tiles + RX_CHUNK - 1) / RX_CHUNK; }

size_t radix_tile_hist_bytes(u64 m)
{
    // sized for the smallest compiled tile so the knob never outgrows a caller's buffer
    const u64 tiles = (m + 3071) / 3072 + 1;
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
// One wave's 64 consecutive digits into its LDS bins.  Sorted and repetitive input (every pass but the first of a sort over text or
// over ranks with many equal values) puts runs of equal digits into neighbouring lanes, and same-address LDS atomics are served one
// lane at a time: a pass over such keys ran at 2.4 TB/s where unsorted keys reach 5.6.  So the lanes of a run elect their first one, and
// it adds the run's length (valid lanes are a prefix of the wave).  The neighbour's digit comes through DPP (row_shr:1 -- no trip
// through the LDS crossbar, which cost the 2-byte sweep over random digits 0.15 ms of 0.45); a row's first lane has no neighbour
// there and starts a run of its own: runs are at most 16 long.
__device__ __forceinline__ void hist_add_runs(u32 *wave_bins, u32 d, bool valid)
{
    const int lane = lane_id();
    const u32 dprev = (u32)__builtin_amdgcn_update_dpp((int)d, (int)d, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    const bool head = valid && ((lane & 15) == 0 || d != dprev);
    const u64 hm = __ballot(head), vm = __ballot(valid);
    if (hm == vm) {                              // (no two neighbours alike -- three waves in four on random digits: nothing to add up)
        if (valid) atomicAdd(&wave_bins[d], 1u);
    } else if (head) {
        const u64 above = lane == 63 ? 0ull : hm >> (lane + 1);
        const u32 next = above ? (u32)lane + (u32)__ffsll((unsigned long long)above) : (u32)__popcll(vm);
        atomicAdd(&wave_bins[d], next - (u32)lane);
    }
}

template <int RX_THREADS, int RX_ITEMS>
__global__ __launch_bounds__(RX_THREADS) void radix_hist_kernel(const u64 *__restrict__ keys, u64 m, int shift,
                                                                 u32 *__restrict__ tile_hist)
{
    constexpr int RX_WAVES = RX_THREADS / 64;
    constexpr int RX_TILE = RX_THREADS * RX_ITEMS;
    __shared__ u32 bins[RX_WAVES][256];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < RX_WAVES * 256; i += RX_THREADS) ((u32 *)bins)[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * RX_TILE;
    // all loads first (one register pair each), then the LDS atomics: keeps 16 loads in flight per lane
    u64 k[RX_ITEMS];
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const u64 i = base + (u64)j * RX_THREADS + tid;
        k[j] = i < m ? keys[i] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const u64 i = base + (u64)j * RX_THREADS + tid;
        hist_add_runs(bins[w], (u32)(k[j] >> shift) & 255u, i < m);
    }
    __syncthreads();
    if (tid < 256) {
        u32 s = 0;
#pragma unroll
        for (int ww = 0; ww < RX_WAVES; ww++) s += bins[ww][tid];
        tile_hist[(u64)blockIdx.x * 256 + tid] = s;
    }
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

// The LDS tile holds the keys first and is then reused for the values (and, optionally, a third
// one-byte stream), the per-wave digit counters are 16-bit and each slot's digit is kept in a byte table, so
// the destination of a slot is re-derived instead of living in a register: 9 B of LDS per element instead
// of 12, two 8192-element tiles fit a CU and one tile's loads overlap the other's ranking.
template <int RX_THREADS, int RX_ITEMS, int MINW, bool HAS_SYM, bool IDENT, bool KEYS_ONLY = false>
__global__ __launch_bounds__(RX_THREADS, MINW) void radix_scatter2_kernel(const u64 *__restrict__ kin, const u32 *__restrict__ vin,
                                                                     u64 *__restrict__ kout, u32 *__restrict__ vout,
                                                                     const u32 *__restrict__ tile_off, u64 m, int shift,
                                                                     const u8 *__restrict__ sin, u8 *__restrict__ sout)
{
    constexpr int RX_WAVES = RX_THREADS / 64;
    constexpr int RX_TILE = RX_THREADS * RX_ITEMS;
    static_assert(RX_TILE <= 65536 && RX_ITEMS % 2 == 0, "positions are packed as 16-bit pairs");
    extern __shared__ __attribute__((aligned(16))) char rx_smem[];
    u64 *stage = (u64 *)rx_smem;                                   // RX_TILE keys, later values, later bytes
    u32 *dbase = (u32 *)(stage + RX_TILE);                         // 256
    u32 *gbase = dbase + 256;                                      // 256
    u32 *scan_sm = gbase + 256;                                    // RX_WAVES (padded to 16)
    u16 (*whist)[256] = (u16 (*)[256])(scan_sm + 16);              // RX_WAVES x 256
    u8 *sdig = (u8 *)(whist + RX_WAVES);                           // RX_TILE: digit of the element in slot s

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u64 tile = rx_tile_of_block((m + RX_TILE - 1) / RX_TILE);
    const u64 tile_base = tile * RX_TILE;
    if (tile_base >= m) return;                                   // padding block of the XCD-aligned grid
    const u64 wave_base = tile_base + (u64)w * (64 * RX_ITEMS);
    const u64 remain = m - tile_base;
    const u32 tile_count = remain < RX_TILE ? (u32)remain : (u32)RX_TILE;

    for (int i = tid; i < RX_WAVES * 128; i += RX_THREADS) ((u32 *)whist)[i] = 0;

    u64 key[RX_ITEMS];
    u32 posp[RX_ITEMS / 2];                                        // two 16-bit tile positions per register
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const u64 i = wave_base + (u64)j * 64 + lane;
        key[j] = i < m ? kin[i] : ~0ull;
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const bool valid = wave_base + (u64)j * 64 + lane < m;
        const u32 d = (u32)(key[j] >> shift) & 255u;
        const u64 peers = match_digit8(d, valid);
        const u32 before = (u32)__popcll(peers & lanemask_lt());
        const u32 cnt = (u32)__popcll(peers);
        const u32 prev = whist[w][d];
        if (j & 1) posp[j >> 1] |= (prev + before) << 16; else posp[j >> 1] = prev + before;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid && before == 0) whist[w][d] = (u16)(prev + cnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();
    {
        u32 run = 0;
        if (tid < 256) {
#pragma unroll
            for (int ww = 0; ww < RX_WAVES; ww++) {
                const u32 c = whist[ww][tid];
                whist[ww][tid] = (u16)run;
                run += c;
            }
        }
        u32 total;
        const u32 exc = block_scan_exclusive<u32, OpAdd, RX_WAVES>(run, OpAdd(), 0u, scan_sm, &total);
        if (tid < 256) {
            dbase[tid] = exc;
            gbase[tid] = tile_off[tile * 256 + tid] - exc;
        }
    }
    __syncthreads();

    // From here on every loop is split into "all LDS reads", then "all dependent work": inside one loop body the
    // compiler keeps a read, its s_waitcnt and the dependent LDS/global operation together, which serialises sixteen
    // LDS round trips per phase (seen in the ISA); batched, a phase costs about one.
#define POS_GET(j) (((j) & 1) ? posp[(j) >> 1] >> 16 : posp[(j) >> 1] & 0xffffu)
#define POS_SET(j, v) do { if ((j) & 1) posp[(j) >> 1] = (posp[(j) >> 1] & 0xffffu) | ((v) << 16); \
                           else posp[(j) >> 1] = (posp[(j) >> 1] & 0xffff0000u) | (v); } while (0)
    constexpr int HB = RX_ITEMS % 8 == 0 ? 8 : 4;                  // LDS reads batched per phase step
#pragma unroll
    for (int j0 = 0; j0 < RX_ITEMS; j0 += HB) {
        u32 add[HB];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 d = (u32)(key[j0 + jj] >> shift) & 255u;
            add[jj] = dbase[d] + whist[w][d];
        }
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const int j = j0 + jj;
            const bool valid = wave_base + (u64)j * 64 + lane < m;
            const u32 p = POS_GET(j) + add[jj];
            POS_SET(j, p);
            if (valid) stage[p] = key[j];
        }
    }
    // the keys' registers are free: the value (and byte) loads overlap the key write-out
    u32 val[KEYS_ONLY ? 1 : RX_ITEMS];
    u32 sym[HAS_SYM ? RX_ITEMS : 1];
    if (!KEYS_ONLY) {
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) {
            const u64 i = wave_base + (u64)j * 64 + lane;
            val[KEYS_ONLY ? 0 : j] = IDENT ? (u32)i : (i < m ? vin[i] : 0u);          // first pass of a sort over positions: value = index
        }
    }
    if (HAS_SYM) {
        // one register per byte and no arithmetic on them here: the 16 loads issue back to back (packing them
        // on arrival made the compiler wait for memory after every two or three loads)
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) {
            const u64 i = wave_base + (u64)j * 64 + lane;
            sym[j] = i < m ? (u32)sin[i] : 0u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j0 = 0; j0 < RX_ITEMS; j0 += HB) {
        u64 k[HB];
        u32 g[HB];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 s = (u32)(j0 + jj) * RX_THREADS + tid;
            k[jj] = s < tile_count ? stage[s] : 0ull;
        }
#pragma unroll
        for (int jj = 0; jj < HB; jj++) g[jj] = gbase[(u32)(k[jj] >> shift) & 255u];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 s = (u32)(j0 + jj) * RX_THREADS + tid;
            if (s < tile_count) {
                sdig[s] = (u8)((u32)(k[jj] >> shift) & 255u);
                kout[g[jj] + s] = k[jj];
            }
        }
    }
    if (KEYS_ONLY) return;                                          // a sort of bare keys (the payload sits in their upper bits)
    __syncthreads();
    // second trip through the same LDS: the value, with the travelling byte in the upper half of the slot
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const bool valid = wave_base + (u64)j * 64 + lane < m;
        const u32 p = POS_GET(j);
        if (valid) {
            if (HAS_SYM) stage[p] = (u64)val[KEYS_ONLY ? 0 : j] | ((u64)sym[j] << 32);
            else ((u32 *)stage)[p] = val[KEYS_ONLY ? 0 : j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j0 = 0; j0 < RX_ITEMS; j0 += HB) {
        u64 e[HAS_SYM ? HB : 1];
        u32 v32[HAS_SYM ? 1 : HB];
        u32 g[HB];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 s = (u32)(j0 + jj) * RX_THREADS + tid;
            const u32 sc = s < tile_count ? s : 0u;
            if (HAS_SYM) e[jj] = stage[sc]; else v32[jj] = ((u32 *)stage)[sc];
            g[jj] = sdig[sc];
        }
#pragma unroll
        for (int jj = 0; jj < HB; jj++) g[jj] = gbase[g[jj]];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 s = (u32)(j0 + jj) * RX_THREADS + tid;
            if (s < tile_count) {
                const u32 dst = g[jj] + s;
                if (HAS_SYM) { vout[dst] = (u32)e[jj]; sout[dst] = (u8)(e[jj] >> 32); }
                else vout[dst] = v32[jj];
            }
        }
    }
#undef POS_GET
#undef POS_SET
}

// ------------------------------------------------------------------------------------
// packed streams (round 0 of the forward transform)
// ------------------------------------------------------------------------------------
// A round-0 key has at most 40 significant bits, yet a pass of the kernel above moves 8 + 4 + 1 bytes per element.
// Between the first and the last pass the element travels instead as
//     lo  u32   key bits 0..31
//     val u32   the position
//     c   u16   key bits 32..39 | carried byte << 8        (HI16)   -- or u8: the carried byte alone (keys <= 32 bits)
// = 10 (9) bytes, and a histogram sweep reads only the stream that holds its digit (4 or 2 bytes, not 8).
// keybuild leaves lo and c in exactly this form (forward.hip, KeyStore), the first pass supplies value = index, and the
// last pass writes wide keys, so everything downstream sees sorted u64 keys.  Ranking, LDS staging and XCD mapping are those of
// radix_scatter2_kernel; the LDS tile carries the digit's stream on its first trip and the other two on its second.
template <typename T>
__global__ __launch_bounds__(512) void radix_hist_packed_kernel(const T *__restrict__ src, u64 m, int shift, u32 *__restrict__ tile_hist)
{
    constexpr int RX_THREADS = 512, RX_ITEMS = 16, RX_WAVES = 8, RX_TILE = 8192;
    __shared__ u32 bins[RX_WAVES][256];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < RX_WAVES * 256; i += RX_THREADS) ((u32 *)bins)[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * RX_TILE;
    T k[RX_ITEMS];
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const u64 i = base + (u64)j * RX_THREADS + tid;
        k[j] = i < m ? src[i] : (T)0;
    }
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const u64 i = base + (u64)j * RX_THREADS + tid;
        hist_add_runs(bins[w], (u32)(k[j] >> shift) & 255u, i < m);
    }
    __syncthreads();
    if (tid < 256) {
        u32 s = 0;
#pragma unroll
        for (int ww = 0; ww < RX_WAVES; ww++) s += bins[ww][tid];
        tile_hist[(u64)blockIdx.x * 256 + tid] = s;
    }
}

// ---- CHAINED passes: no per-tile histogram sweep, no column scan --------------------------------------------------------------
// A pass needs, per tile and digit, the number of elements with that digit in all EARLIER tiles.  The classic pass gets it from a
// histogram sweep over the pass's input (4 or 2 bytes per element) and a column scan of the [tile][digit] table; five such sweeps
// cost 3.6 of a 35.6 ms forward at n = 2^30.  The chained pass computes it while it runs (decoupled look-back, Merrill & Garland):
// the digit totals of ALL passes come from ONE sweep over the keys before the first pass (the multiset of keys never changes), a
// tile publishes its own digit counts as soon as it has ranked its elements (AGGREGATE), adds up its predecessors' counts walking
// backwards until it meets one that already knows its inclusive prefix, and publishes its own inclusive prefix (INCLUSIVE).
//   state word (u64, one per tile and digit): flag (0 nothing, 1 aggregate, 2 inclusive) << 62 | pass tag (1..7) << 59 | value (40 bits);
//   cleared once per sort, the tag tells the passes apart.
// Order and progress: a workgroup takes a TICKET when it starts (one counter per pass) and the ticket picks the tile, so a tile only
// ever waits for tiles whose workgroups have started or are among the next 63 to start -- 64 workgroups are always resident
// together (checked against the occupancy API once per context), so every wait ends; every spin is bounded all the same and a
// bound that is hit raises an error word instead of hanging the GPU.  Inside a group of 64 consecutive tickets the tiles are dealt so
// that ticket t (XCD t % 8 when workgroups start in dispatch order) gets eight CONSECUTIVE tiles' worth of neighbours: seam
// cache lines between consecutive tiles' runs are still written through one L2 (rx_tile_of_block's reason), seven times out of eight.
#define RXC_FLAG_AGG   1ull
#define RXC_FLAG_INCL  2ull
#define RXC_VALUE_MASK ((1ull << 40) - 1ull)
#define RXC_SPIN_MAX   (1u << 20)
#define RXC_GROUP      64u
#define RXC_RUN        8u
struct ChainIO {
    u64 *state;                  // [tiles][256]
    unsigned int *ticket;        // [8] one per pass; [7] = error word
    const u64 *gbase;            // [passes][256] exclusive digit totals of the pass
    u32 tag;                     // 1 + pass
};
__device__ __forceinline__ u64 rxc_word(u64 flag, u32 tag, u64 value) { return (flag << 62) | ((u64)tag << 59) | value; }

// digit totals of all passes in one sweep: keys as keybuild leaves them (lo: bits 0..31, c: key bits 32..39 in the low byte when HI16)
template <bool HI16>
__global__ __launch_bounds__(512) void radix_global_hist_kernel(const u32 *__restrict__ lo, const u16 *__restrict__ c, u64 m, int passes, unsigned long long *__restrict__ gh)
{
    constexpr int NP = 5;
    __shared__ u32 bins[NP][256];
    const int tid = threadIdx.x;
    for (int i = tid; i < NP * 256; i += 512) ((u32 *)bins)[i] = 0;
    __syncthreads();
    const u64 per = 512ull * 8;
    for (u64 base = (u64)blockIdx.x * per; base < m; base += (u64)gridDim.x * per) {
        u32 k[8]; u32 h[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u64 i = base + (u64)j * 512 + tid;
            k[j] = i < m ? lo[i] : 0u;
            h[j] = (HI16 && i < m) ? (u32)c[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u64 i = base + (u64)j * 512 + tid;
            if (i < m) {
                atomicAdd(&bins[0][k[j] & 255u], 1u);
                atomicAdd(&bins[1][(k[j] >> 8) & 255u], 1u);
                atomicAdd(&bins[2][(k[j] >> 16) & 255u], 1u);
                if (passes > 3) atomicAdd(&bins[3][k[j] >> 24], 1u);
                if (HI16) atomicAdd(&bins[4][h[j] & 255u], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < passes * 256; i += 512) { const u32 v = ((u32 *)bins)[i]; if (v) atomicAdd(&gh[i], (unsigned long long)v); }
}
// gh[p][d] -> exclusive sums over d, in place (one workgroup of 256 threads per pass)
__global__ __launch_bounds__(256) void radix_global_base_kernel(unsigned long long *__restrict__ gh)
{
    __shared__ u64 sm[4];
    const int d = threadIdx.x;
    unsigned long long *row = gh + (size_t)blockIdx.x * 256;
    const u64 v = row[d];
    u64 tot;
    const u64 exc = block_scan_exclusive<u64, OpAdd, 4>(v, OpAdd(), (u64)0, sm, &tot);
    row[d] = exc;
}

struct PackedIO {
    const u32 *lo_in; const void *c_in;                    // c: u16 if HI16 else u8
    const u32 *val_in;                                     // not read by the first pass (value = index)
    u64 *kout_wide; u8 *sym_out;                           // OUT_WIDE
    u32 *lo_out; void *c_out;                              // !OUT_WIDE
    u32 *val_out;
};

template <bool FIRST, bool OUT_WIDE, bool HI16, bool CHAIN = false>
__global__ __launch_bounds__(512, 4) void radix_scatter_packed_kernel(PackedIO io, const u32 *__restrict__ tile_off, u64 m, int shift, ChainIO ch)
{
    constexpr int RX_THREADS = 512, RX_ITEMS = 16, RX_WAVES = 8, RX_TILE = 8192;
    extern __shared__ __attribute__((aligned(16))) char rx_smem[];
    u64 *stage = (u64 *)rx_smem;                                   // RX_TILE x u32 (first trip), RX_TILE x uint2 (second)
    u32 *dbase = (u32 *)(stage + RX_TILE);                         // 256
    u32 *gbase = dbase + 256;                                      // 256
    u32 *scan_sm = gbase + 256;                                    // RX_WAVES (padded to 16)
    u16 (*whist)[256] = (u16 (*)[256])(scan_sm + 16);              // RX_WAVES x 256
    u8 *sdig = (u8 *)(whist + RX_WAVES);                           // RX_TILE: digit of the element in slot s

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    u64 tile;
    if (CHAIN) {
        // the ticket picks the tile (see CHAINED passes): tickets of a group of 64 are dealt 8 x 8
        __shared__ u32 s_ticket;
        if (tid == 0) s_ticket = atomicAdd(&ch.ticket[ch.tag - 1u], 1u);
        __syncthreads();
        const u32 b = s_ticket, r = b % RXC_GROUP;
        tile = (u64)(b - r) + (u64)(r % RXC_RUN) * (RXC_GROUP / RXC_RUN) + (u64)(r / RXC_RUN);
    } else tile = rx_tile_of_block((m + RX_TILE - 1) / RX_TILE);
    const u64 tile_base = tile * RX_TILE;
    if (tile_base >= m) return;                                   // padding block of the grid
    const u64 wave_base = tile_base + (u64)w * (64 * RX_ITEMS);
    const u64 remain = m - tile_base;
    const u32 tile_count = remain < RX_TILE ? (u32)remain : (u32)RX_TILE;

    for (int i = tid; i < RX_WAVES * 128; i += RX_THREADS) ((u32 *)whist)[i] = 0;

    // element j of this lane is i0 + 64 j; which of the sixteen exist is decided once (bit j of vmask), so that no
    // 64-bit index has to stay alive for the bounds checks further down
    const u64 i0 = wave_base + (u64)lane;
    u32 vmask = 0;
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) vmask |= (i0 + (u64)j * 64 < m ? 1u : 0u) << j;
#define PK_VALID(j) ((vmask >> (j)) & 1u)

    // Only the stream that holds this pass's digit is loaded before the ranking and makes the first LDS trip alone;
    // the other two are loaded while the first trip is written out and share the second trip.
    // With 40-bit keys the fifth (= last) pass sorts by the byte that travels in c.
    constexpr bool DIG_C = HI16 && OUT_WIDE;
    u32 dsrc[RX_ITEMS];
    u32 posp[RX_ITEMS / 2];
    {
        const u16 *c16 = (const u16 *)io.c_in + i0;
        const u32 *lop = io.lo_in + i0;
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) {
            if (DIG_C) dsrc[j] = PK_VALID(j) ? (u32)c16[j * 64] : 0xffffu;
            else dsrc[j] = PK_VALID(j) ? lop[j * 64] : ~0u;
        }
    }
#define PK_DIGIT(j) (DIG_C ? (dsrc[j] & 255u) : ((dsrc[j] >> shift) & 255u))
    __syncthreads();

#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const bool valid = PK_VALID(j);
        const u32 d = PK_DIGIT(j);
        const u64 peers = match_digit8(d, valid);
        const u32 before = (u32)__popcll(peers & lanemask_lt());
        const u32 cnt = (u32)__popcll(peers);
        const u32 prev = whist[w][d];
        if (j & 1) posp[j >> 1] |= (prev + before) << 16; else posp[j >> 1] = prev + before;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid && before == 0) whist[w][d] = (u16)(prev + cnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();
    u32 crun = 0, cexc = 0;              // (CHAIN) this tile's count of digit `tid` and the tile-local exclusive prefix over the digits
    {
        u32 run = 0;
        if (tid < 256) {
#pragma unroll
            for (int ww = 0; ww < RX_WAVES; ww++) {
                const u32 cc = whist[ww][tid];
                whist[ww][tid] = (u16)run;
                run += cc;
            }
        }
        u32 total;
        const u32 exc = block_scan_exclusive<u32, OpAdd, RX_WAVES>(run, OpAdd(), 0u, scan_sm, &total);
        if (tid < 256) {
            dbase[tid] = exc;
            if (CHAIN) {
                // this tile's count of digit `tid`: out at once (the first tile knows its inclusive prefix already); the look-back
                // itself waits until the first LDS trip has been staged and the other streams' loads are in flight (below)
                crun = run; cexc = exc;
                __hip_atomic_store(&ch.state[tile * 256 + tid], rxc_word(tile == 0 ? RXC_FLAG_INCL : RXC_FLAG_AGG, ch.tag, (u64)run), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            } else gbase[tid] = tile_off[tile * 256 + tid] - exc;
        }
    }
    __syncthreads();

#define POS_GET(j) (((j) & 1) ? posp[(j) >> 1] >> 16 : posp[(j) >> 1] & 0xffffu)
#define POS_SET(j, v) do { if ((j) & 1) posp[(j) >> 1] = (posp[(j) >> 1] & 0xffffu) | ((v) << 16); \
                           else posp[(j) >> 1] = (posp[(j) >> 1] & 0xffff0000u) | (v); } while (0)
    constexpr int HB = 8;
    u32 *stage32 = (u32 *)stage;
    // first trip: the digit's stream into its slot, the slot's digit into the byte table
#pragma unroll
    for (int j0 = 0; j0 < RX_ITEMS; j0 += HB) {
        u32 add[HB];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 d = PK_DIGIT(j0 + jj);
            add[jj] = dbase[d] + whist[w][d];
        }
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const int j = j0 + jj;
            const bool valid = PK_VALID(j);
            const u32 p = POS_GET(j) + add[jj];
            POS_SET(j, p);
            if (valid) { stage32[p] = dsrc[j]; sdig[p] = (u8)PK_DIGIT(j); }
        }
    }
    // the other two streams: (lo, val) on the pass that sorts by c, else (val, c); their loads overlap the write-out below
    u32 sa[RX_ITEMS], sb[RX_ITEMS];
    {
        const u32 *lop = io.lo_in + i0, *valp = io.val_in + i0;
        const u8 *c8 = (const u8 *)io.c_in + i0;
        const u16 *c16 = (const u16 *)io.c_in + i0;
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) {
            if (FIRST) sa[j] = 0u;                                   // value = index, formed when it is staged
            else if (DIG_C) sa[j] = PK_VALID(j) ? lop[j * 64] : 0u;
            else sa[j] = PK_VALID(j) ? valp[j * 64] : 0u;
        }
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) {
            if (DIG_C) sb[j] = PK_VALID(j) ? valp[j * 64] : 0u;
            else sb[j] = PK_VALID(j) ? (HI16 ? (u32)c16[j * 64] : (u32)c8[j * 64]) : 0u;
        }
    }
    if (CHAIN && tid < 256) {
        // look-back: thread `tid` adds up digit tid's counts of the tiles before this one, nearest first, four state words in flight
        u64 prefix = 0;
        if (tile != 0) {
            u64 p = tile;                                    // tiles p - 1, p - 2, ... are looked at next
            bool done = false;
            u32 spins = 0;
            while (!done) {
                u64 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    v[q] = p > (u64)q ? __hip_atomic_load(&ch.state[(p - 1 - (u64)q) * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                int used = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (done || used != q) continue;
                    if (p <= (u64)q) { done = true; continue; }          // (cannot happen: tile 0 is inclusive)
                    const u64 x = v[q];
                    if (((u32)(x >> 59) & 7u) != ch.tag || (x >> 62) == 0ull) continue;      // not there yet: look again from here
                    prefix += x & RXC_VALUE_MASK;
                    used = q + 1;
                    if ((x >> 62) == RXC_FLAG_INCL) done = true;
                }
                p -= (u64)used;
                if (!done && used == 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > RXC_SPIN_MAX) { __hip_atomic_store(&ch.ticket[7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); done = true; }
                }
            }
            __hip_atomic_store(&ch.state[tile * 256 + tid], rxc_word(RXC_FLAG_INCL, ch.tag, prefix + (u64)crun), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gbase[tid] = (u32)(ch.gbase[(ch.tag - 1u) * 256 + tid] + prefix) - cexc;
    }
    __syncthreads();
    u32 hi_s[DIG_C ? RX_ITEMS : 1];                                // last 40-bit pass: the slot's key byte waits for its lo
#pragma unroll
    for (int j0 = 0; j0 < RX_ITEMS; j0 += HB) {
        u32 x[HB], g[HB];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 s = (u32)(j0 + jj) * RX_THREADS + tid;
            const u32 sc = s < tile_count ? s : 0u;
            x[jj] = stage32[sc];
            g[jj] = sdig[sc];
        }
#pragma unroll
        for (int jj = 0; jj < HB; jj++) g[jj] = gbase[g[jj]];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 s = (u32)(j0 + jj) * RX_THREADS + tid;
            if (DIG_C) hi_s[j0 + jj] = x[jj] & 255u;
            if (s < tile_count) {
                const u32 dst = g[jj] + s;
                if (DIG_C) io.sym_out[dst] = (u8)(x[jj] >> 8);
                else if (OUT_WIDE) io.kout_wide[dst] = (u64)x[jj];
                else io.lo_out[dst] = x[jj];
            }
        }
    }
    __syncthreads();
    // second trip: the pair.  (The index value is rebuilt from an opaque copy of the base: left to itself the compiler
    // keeps the sixteen indices of the load addresses alive from the top of the kernel and spills them.)
    u32 idx_base = (u32)wave_base + (u32)lane;
    asm volatile("" : "+v"(idx_base));
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const bool valid = PK_VALID(j);
        const u32 second = sb[j];
        const u32 firstw = FIRST ? idx_base + (u32)j * 64u : sa[j];
        if (valid) ((uint2 *)stage)[POS_GET(j)] = make_uint2(firstw, second);
    }
    __syncthreads();
#pragma unroll
    for (int j0 = 0; j0 < RX_ITEMS; j0 += HB) {
        uint2 e[HB];
        u32 g[HB];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 s = (u32)(j0 + jj) * RX_THREADS + tid;
            const u32 sc = s < tile_count ? s : 0u;
            e[jj] = ((const uint2 *)stage)[sc];
            g[jj] = sdig[sc];
        }
#pragma unroll
        for (int jj = 0; jj < HB; jj++) g[jj] = gbase[g[jj]];
#pragma unroll
        for (int jj = 0; jj < HB; jj++) {
            const u32 s = (u32)(j0 + jj) * RX_THREADS + tid;
            if (s < tile_count) {
                const u32 dst = g[jj] + s;
                if (DIG_C) {
                    io.kout_wide[dst] = (u64)e[jj].x | ((u64)hi_s[j0 + jj] << 32);
                    io.val_out[dst] = e[jj].y;
                } else {
                    io.val_out[dst] = e[jj].x;
                    if (OUT_WIDE) io.sym_out[dst] = (u8)e[jj].y;
                    else if (HI16) ((u16 *)io.c_out)[dst] = (u16)e[jj].y;
                    else ((u8 *)io.c_out)[dst] = (u8)e[jj].y;
                }
            }
        }
    }
#undef POS_GET
#undef POS_SET
#undef PK_DIGIT
#undef PK_VALID
}

#define RX_PACKED_LDS ((size_t)8192 * 9 + 2048 + 64 + (size_t)8 * 512)
template <bool FIRST, bool OUT_WIDE, bool HI16>
static int launch_scatter_packed(bwts_ctx *ctx, u64 tiles, const PackedIO &io, const u32 *tile_off, u64 m, int shift, const ChainIO *ch = nullptr)
{
    constexpr size_t lds = RX_PACKED_LDS;
    if (ch) {
        BWTS_TRY(ensure_dyn_lds(ctx, (const void *)radix_scatter_packed_kernel<FIRST, OUT_WIDE, HI16, true>, lds));
        const u64 grid = (tiles + RXC_GROUP - 1) / RXC_GROUP * RXC_GROUP;
        radix_scatter_packed_kernel<FIRST, OUT_WIDE, HI16, true><<<dim3((unsigned)grid), dim3(512), lds, ctx->stream>>>(io, tile_off, m, shift, *ch);
        return BWTS_OK;
    }
    BWTS_TRY(ensure_dyn_lds(ctx, (const void *)radix_scatter_packed_kernel<FIRST, OUT_WIDE, HI16>, lds));
    radix_scatter_packed_kernel<FIRST, OUT_WIDE, HI16><<<dim3((unsigned)rx_grid(tiles)), dim3(512), lds, ctx->stream>>>(io, tile_off, m, shift, ChainIO{});
    return BWTS_OK;
}

// may the chained passes be used on this context?  (BWTS_RX_CHAIN=1 asks for them; 64 workgroups of the pass kernel must be resident together -- with a margin)
static bool radix_chain_ok(bwts_ctx *ctx, u64 m)
{
    if (m < (1ull << 22)) return false;             // (small sorts: the classic pass's table work is nothing, and the state words need the room)
    // (opt-in: exact, but measured slower -- 7.5 against 4.65 ms per pass at n = 2^30, profiles/history/r04_chained_passes.md)
    { const char *e = bwts_knob(ctx, "BWTS_RX_CHAIN"); if (!(e && atoi(e) == 1)) return false; }
    if (ctx->chain_cap == 0) {
        int per_cu = 0, cus = 0;
        ctx->chain_cap = -1;
        if (hipFuncSetAttribute((const void *)radix_scatter_packed_kernel<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RX_PACKED_LDS) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, radix_scatter_packed_kernel<false, false, true, true>, 512, RX_PACKED_LDS) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && per_cu > 0 && cus > 0)
            ctx->chain_cap = per_cu * cus;
        else (void)hipGetLastError();
    }
    return ctx->chain_cap >= 4 * (int)RXC_GROUP;
}

// round 0 with the byte stream and identity values, keys of 17..40 bits: split -> packed ... packed -> wide
template <bool HI16>
static int radix_sort_packed(bwts_ctx *ctx, const SortPlan &plan, u64 m, int passes, int *result_buf)
{
    const u64 tiles = (m + 8191) / 8192;
    u32 *tile_hist = plan.tile_hist;
    const size_t lo_bytes = align_up((size_t)m * 4, 256);
    const u64 pass_bytes = HI16 ? 20 : 18;
    int cur = 0;
    // chained passes (see CHAINED passes above): the state words live where the tile table would, the digit totals and tickets behind them
    const bool chain = radix_chain_ok(ctx, m);
    ChainIO ch{};
    if (chain) {
        const size_t st_bytes = align_up((size_t)tiles * 256 * sizeof(u64), 256);
        if (st_bytes + 8 * 256 * sizeof(u64) + 256 > radix_tile_hist_bytes(m)) return BWTS_E_INTERNAL;
        ch.state = (u64 *)tile_hist;
        unsigned long long *gh = (unsigned long long *)((char *)tile_hist + st_bytes);
        ch.gbase = (const u64 *)gh;
        ch.ticket = (unsigned int *)(gh + 8 * 256);
        HIPC(hipMemsetAsync(tile_hist, 0, st_bytes + 8 * 256 * sizeof(u64) + 256, ctx->stream));
        SpanGuard g(ctx, BWTS_K_RADIX_HIST, m, (HI16 ? 6 : 4) * m);
        radix_global_hist_kernel<HI16><<<dim3(2048), dim3(512), 0, ctx->stream>>>((const u32 *)plan.keys[0], (const u16 *)((char *)plan.keys[0] + lo_bytes), m, passes, gh);
        radix_global_base_kernel<<<dim3((unsigned)passes), dim3(256), 0, ctx->stream>>>(gh);
        HIPC(hipGetLastError());
    }
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        const bool first = p == 0, last = p == passes - 1;
        char *src = (char *)plan.keys[cur], *dst = (char *)plan.keys[cur ^ 1];
        PackedIO io{};
        io.lo_in = (const u32 *)src; io.c_in = src + lo_bytes; io.val_in = plan.vals[cur];
        io.kout_wide = plan.keys[cur ^ 1]; io.sym_out = plan.sym_final;
        io.lo_out = (u32 *)dst; io.c_out = dst + lo_bytes; io.val_out = plan.vals[cur ^ 1];
        ch.tag = (u32)p + 1u;
        const ChainIO *chp = chain ? &ch : nullptr;
        if (!chain) {
            SpanGuard g(ctx, BWTS_K_RADIX_HIST, m, ((HI16 && shift >= 32) ? 2 : 4) * m);
            if (HI16 && shift >= 32) radix_hist_packed_kernel<u16><<<dim3((unsigned)tiles), dim3(512), 0, ctx->stream>>>((const u16 *)io.c_in, m, 0, tile_hist);
            else radix_hist_packed_kernel<u32><<<dim3((unsigned)tiles), dim3(512), 0, ctx->stream>>>(io.lo_in, m, shift, tile_hist);
        }
        if (!chain) {
            SpanGuard g(ctx, BWTS_K_RADIX_SCAN, tiles * 256, tiles * 256 * 12);
            STAGE("packed pass: histogram");
            BWTS_TRY(radix_column_scan(ctx, tile_hist, tiles, plan.scan_temp));
            STAGE("packed pass: column scan");
        }
        if (first) {
            SpanGuard g(ctx, BWTS_K_RADIX_SCATTER, m, (pass_bytes / 2 - 4 + pass_bytes / 2) * m);
            BWTS_TRY((launch_scatter_packed<true, false, HI16>(ctx, tiles, io, tile_hist, m, shift, chp)));
        } else if (last) {
            SpanGuard g(ctx, BWTS_K_RADIX_SCATTER, m, (pass_bytes / 2 + 13) * m);
            BWTS_TRY((launch_scatter_packed<false, true, HI16>(ctx, tiles, io, tile_hist, m, shift, chp)));
        } else {
            // timed a second time under its own class: the roofline kernel (one template variant, n-sized launches)
            SpanGuard g(ctx, BWTS_K_RADIX_SCATTER, m, pass_bytes * m);
            SpanGuard gm(ctx, BWTS_K_RADIX_SCATTER_MAIN, m, pass_bytes * m);
            BWTS_TRY((launch_scatter_packed<false, false, HI16>(ctx, tiles, io, tile_hist, m, shift, chp)));
        }
        HIPC(hipGetLastError());
        STAGE("packed pass: scatter");
        cur ^= 1;
    }
    if (chain) {
        // a look-back that ran into its spin bound has raised the error word: the sort's output is not to be trusted
        unsigned int err = 0;
        HIPC(hipMemcpyAsync(&err, ch.ticket + 7, sizeof(err), hipMemcpyDeviceToHost, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));
        if (err) return BWTS_E_INTERNAL;
    }
    *result_buf = cur;
    return BWTS_OK;
}

// ------------------------------------------------------------------------------------
// small sorts: every pass inside one workgroup
// ------------------------------------------------------------------------------------
// The late rounds of the doubling sort often hold a few hundred stubborn ties; at 7 launches per pass and 7-8 passes per
// sort, launch latency is all they cost.  Up to RS_MAX pairs are sorted here in one launch: the pairs live in LDS
// (ping-pong), a pass ranks with the same ballot matching as the big kernels, a thread owns RS_ITEMS consecutive-by-wave
// elements, and the per-wave digit counts are scanned by the first 256 threads.
#define RS_THREADS 1024
#define RS_WAVES   (RS_THREADS / 64)
#define RS_ITEMS   4
#define RS_MAX     (RS_THREADS * RS_ITEMS)
__global__ __launch_bounds__(RS_THREADS) void radix_sort_small_kernel(const u64 *__restrict__ kin, const u32 *__restrict__ vin,
                                                                      u64 *__restrict__ kout, u32 *__restrict__ vout, u32 m, int passes)
{
    __shared__ u64 sk[2][RS_MAX];
    __shared__ u32 sv[2][RS_MAX];
    __shared__ u16 whist[RS_WAVES][256];
    __shared__ u32 dbase[256];
    __shared__ u32 scan_sm[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (u32 i = tid; i < m; i += RS_THREADS) { sk[0][i] = kin[i]; sv[0][i] = vin[i]; }
    int cur = 0;
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        for (int i = tid; i < RS_WAVES * 128; i += RS_THREADS) ((u32 *)whist)[i] = 0;
        __syncthreads();
        // wave w owns elements [w * 64 * RS_ITEMS, ...), item j of a lane is element base + j * 64 + lane
        const u32 wbase = (u32)w * (64 * RS_ITEMS);
        u64 key[RS_ITEMS];
        u32 val[RS_ITEMS], pos[RS_ITEMS];
#pragma unroll
        for (int j = 0; j < RS_ITEMS; j++) {
            const u32 i = wbase + (u32)j * 64 + lane;
            key[j] = i < m ? sk[cur][i] : ~0ull;
            val[j] = i < m ? sv[cur][i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < RS_ITEMS; j++) {
            const bool valid = wbase + (u32)j * 64 + lane < m;
            const u32 d = (u32)(key[j] >> shift) & 255u;
            const u64 peers = match_digit8(d, valid);
            const u32 before = (u32)__popcll(peers & lanemask_lt());
            const u32 cnt = (u32)__popcll(peers);
            const u32 prev = whist[w][d];
            pos[j] = prev + before;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (valid && before == 0) whist[w][d] = (u16)(prev + cnt);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        __syncthreads();
        {
            u32 run = 0;
            if (tid < 256) {
#pragma unroll
                for (int ww = 0; ww < RS_WAVES; ww++) {
                    const u32 c = whist[ww][tid];
                    whist[ww][tid] = (u16)run;
                    run += c;
                }
            }
            u32 total;
            const u32 exc = block_scan_exclusive<u32, OpAdd, RS_WAVES>(run, OpAdd(), 0u, scan_sm, &total);
            if (tid < 256) dbase[tid] = exc;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; j++) {
            const bool valid = wbase + (u32)j * 64 + lane < m;
            const u32 d = (u32)(key[j] >> shift) & 255u;
            if (valid) {
                const u32 dst = dbase[d] + whist[w][d] + pos[j];
                sk[cur ^ 1][dst] = key[j];
                sv[cur ^ 1][dst] = val[j];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    for (u32 i = tid; i < m; i += RS_THREADS) { kout[i] = sk[cur][i]; vout[i] = sv[cur][i]; }
}

// ------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------
template <int TH, int IT>
static void launch_hist_t(bwts_ctx *ctx, u64 tiles, const u64 *keys, u64 m, int shift, u32 *tile_hist)
{
    radix_hist_kernel<TH, IT><<<dim3((unsigned)tiles), dim3(TH), 0, ctx->stream>>>(keys, m, shift, tile_hist);
}

template <int TH, int IT, int MINW, bool HAS_SYM = false, bool IDENT = false>
static int launch_scatter2_t(bwts_ctx *ctx, u64 tiles, const u64 *kin, const u32 *vin, u64 *kout, u32 *vout, const u32 *tile_off,
                             u64 m, int shift, const u8 *sin = nullptr, u8 *sout = nullptr)
{
    constexpr size_t lds = (size_t)TH * IT * 9 + 2048 + 64 + (size_t)(TH / 64) * 512;
    BWTS_TRY(ensure_dyn_lds(ctx, (const void *)radix_scatter2_kernel<TH, IT, MINW, HAS_SYM, IDENT>, lds));
    radix_scatter2_kernel<TH, IT, MINW, HAS_SYM, IDENT><<<dim3((unsigned)rx_grid(tiles)), dim3(TH), lds, ctx->stream>>>(kin, vin, kout, vout, tile_off, m, shift, sin, sout);
    return BWTS_OK;
}

static void launch_hist(bwts_ctx *ctx, int cfg, u64 tiles, const u64 *keys, u64 m, int shift, u32 *tile_hist)
{
    switch (cfg) {
    case 0: launch_hist_t<512, 16>(ctx, tiles, keys, m, shift, tile_hist); break;
    case 1: launch_hist_t<512, 12>(ctx, tiles, keys, m, shift, tile_hist); break;
    case 2: launch_hist_t<256, 16>(ctx, tiles, keys, m, shift, tile_hist); break;
    default: launch_hist_t<1024, 8>(ctx, tiles, keys, m, shift, tile_hist); break;
    }
}

static int launch_scatter(bwts_ctx *ctx, int cfg, u64 tiles, const u64 *kin, const u32 *vin, u64 *kout, u32 *vout,
                          const u32 *tile_off, u64 m, int shift)
{
    switch (cfg) {
    case 0: return launch_scatter2_t<512, 16, 4>(ctx, tiles, kin, vin, kout, vout, tile_off, m, shift);
    case 1: return launch_scatter2_t<512, 12, 4>(ctx, tiles, kin, vin, kout, vout, tile_off, m, shift);
    case 2: return launch_scatter2_t<256, 16, 4>(ctx, tiles, kin, vin, kout, vout, tile_off, m, shift);
    default: return launch_scatter2_t<1024, 8, 8>(ctx, tiles, kin, vin, kout, vout, tile_off, m, shift);
    }
}

// [tile][digit] counts -> global exclusive offsets in digit-major order, in place.
// tile_hist must be followed by the chunk table (radix_tile_hist_bytes()).
// The three steps above in one launch, for tables of at most RX_FUSED_CHUNKS chunks (sorts of up to 2^28 elements): one workgroup
// per chunk, thread = digit.  Every workgroup publishes its chunk's column sums, waits until all have (at most 256 workgroups of
// 256 threads: always resident together, and the only wait in the kernel), then adds up what lies before its chunk -- the chunks
// before it in its own column, all of the smaller digits' columns -- and rewrites its chunk.  sync[0] counts arrivals, sync[1]
// departures; the last workgroup to leave puts both back to zero for the next launch.
// (The sort of a few million pairs spent as long in the five launches it replaces as in the histogram sweep itself.)
#define RX_FUSED_CHUNKS 256
__global__ __launch_bounds__(256) void radix_column_scan_fused_kernel(u32 *__restrict__ tile_hist, u64 tiles, u32 chunks, u32 *__restrict__ chunk_sum,
                                                                      unsigned int *__restrict__ sync)
{
    __shared__ u32 scan_sm[4];
    const u32 c = blockIdx.x, d = threadIdx.x;
    const u64 t0 = (u64)c * RX_CHUNK;
    const u64 t1 = t0 + RX_CHUNK < tiles ? t0 + RX_CHUNK : tiles;
    // (both sweeps over the chunk's tiles keep eight loads in flight: with one at a time a chunk of 128 tiles took 16 us)
    u32 s = 0;
    {
        u64 t = t0;
        for (; t + 8 <= t1; t += 8) {
            u32 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = tile_hist[(t + j) * 256 + d];
#pragma unroll
            for (int j = 0; j < 8; j++) s += v[j];
        }
        for (; t < t1; t++) s += tile_hist[t * 256 + d];
    }
    u32 before = 0, total = s;
    if (chunks > 1) {            // (one chunk -- up to 2^20 elements: nobody to wait for)
        chunk_sum[(u64)c * 256 + d] = s;
        __threadfence();                         // every thread's sums are out before the workgroup reports in
        __syncthreads();
        if (d == 0) {
            __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&sync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < chunks) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __threadfence();                         // acquire for the whole workgroup: the plain loads below see the other workgroups' sums
        total = 0;
        u32 cc = 0;
        for (; cc + 4 <= chunks; cc += 4) {      // independent loads, four in flight
            const u32 v0 = chunk_sum[(u64)cc * 256 + d], v1 = chunk_sum[(u64)(cc + 1) * 256 + d];
            const u32 v2 = chunk_sum[(u64)(cc + 2) * 256 + d], v3 = chunk_sum[(u64)(cc + 3) * 256 + d];
            total += v0 + v1 + v2 + v3;
            before += (cc < c ? v0 : 0u) + (cc + 1 < c ? v1 : 0u) + (cc + 2 < c ? v2 : 0u) + (cc + 3 < c ? v3 : 0u);
        }
        for (; cc < chunks; cc++) { const u32 v = chunk_sum[(u64)cc * 256 + d]; total += v; before += cc < c ? v : 0u; }
    }
    u32 all;
    const u32 base = block_scan_exclusive<u32, OpAdd, 4>(total, OpAdd(), 0u, scan_sm, &all);      // elements with a smaller digit
    u32 run = base + before;
    {
        u64 t = t0;
        for (; t + 8 <= t1; t += 8) {
            u32 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = tile_hist[(t + j) * 256 + d];
#pragma unroll
            for (int j = 0; j < 8; j++) { tile_hist[(t + j) * 256 + d] = run; run += v[j]; }
        }
        for (; t < t1; t++) { const u32 v = tile_hist[t * 256 + d]; tile_hist[t * 256 + d] = run; run += v; }
    }
    if (chunks > 1) {
        __syncthreads();
        if (d == 0) {
            const u32 left = __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (left + 1 == chunks) {          // everyone has read the sums and passed the wait
                __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int radix_column_scan(bwts_ctx *ctx, u32 *tile_hist, u64 tiles, void *scan_temp)
{
    const u64 chunks = radix_chunks(tiles);
    u32 *chunk_sum = (u32 *)((char *)tile_hist + align_up((size_t)tiles * 256 * sizeof(u32), 256));
    const bool fused_ok = [ctx] { const char *e = bwts_knob(ctx, "BWTS_RX_FUSED_SCAN"); return !(e && atoi(e) == 0); }();
    if (ctx->fused_scan_cap == 0) {
        // every workgroup of the fused kernel waits for all the others: it may only be used with as many workgroups as the device
        // is certain to hold at once (a whole MI355X: 256 CUs x 8; a partition of it, or a CU-masked queue, fewer)
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, radix_column_scan_fused_kernel, 256, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        ctx->fused_scan_cap = per_cu > 0 && cus > 0 ? per_cu * cus : -1;
    }
    if (fused_ok && chunks <= RX_FUSED_CHUNKS && (long long)chunks <= (long long)ctx->fused_scan_cap) {
        radix_column_scan_fused_kernel<<<dim3((unsigned)chunks), dim3(256), 0, ctx->stream>>>(tile_hist, tiles, (u32)chunks, chunk_sum,
                                                                                             (unsigned int *)(ctx->d_small + SM_RX_SYNC));
        HIPC(hipGetLastError());
        return BWTS_OK;
    }
    radix_chunk_sum_kernel<<<dim3((unsigned)chunks), dim3(256), 0, ctx->stream>>>(tile_hist, tiles, chunks, chunk_sum);
    BWTS_TRY(exclusive_sum_u32(ctx, chunk_sum, chunks * 256, scan_temp));
    radix_chunk_apply_kernel<<<dim3((unsigned)chunks), dim3(256), 0, ctx->stream>>>(tile_hist, tiles, chunks, chunk_sum);
    HIPC(hipGetLastError());
    return BWTS_OK;
}

// Stable sort of bare u64 keys on bits [lo_bit, lo_bit + bits): 16 bytes per element and pass.  Product tile shape only.
// Returns in *result_buf which of keys[] holds the output.
int radix_sort_keys(bwts_ctx *ctx, u64 *keys[2], u32 *tile_hist, void *scan_temp, u64 m, int lo_bit, int bits, int *result_buf)
{
    int cur = 0;
    const u64 tiles = (m + 8191) / 8192;
    constexpr size_t lds = (size_t)512 * 16 * 9 + 2048 + 64 + (size_t)8 * 512;
    BWTS_TRY(ensure_dyn_lds(ctx, (const void *)radix_scatter2_kernel<512, 16, 4, false, false, true>, lds));
    for (int shift = lo_bit; shift < lo_bit + bits; shift += 8) {
        {
            SpanGuard g(ctx, BWTS_K_RADIX_HIST, m, 8 * m);
            launch_hist_t<512, 16>(ctx, tiles, keys[cur], m, shift, tile_hist);
        }
        {
            SpanGuard g(ctx, BWTS_K_RADIX_SCAN, tiles * 256, tiles * 256 * 12);
            BWTS_TRY(radix_column_scan(ctx, tile_hist, tiles, scan_temp));
        }
        SpanGuard g(ctx, BWTS_K_RADIX_SCATTER, m, 16 * m);
        radix_scatter2_kernel<512, 16, 4, false, false, true><<<dim3((unsigned)rx_grid(tiles)), dim3(512), lds, ctx->stream>>>(
            keys[cur], nullptr, keys[cur ^ 1], nullptr, tile_hist, m, shift, nullptr, nullptr);
        HIPC(hipGetLastError());
        cur ^= 1;
    }
    *result_buf = cur;
    return BWTS_OK;
}

bool radix_supports_sym(const bwts_ctx *ctx) { return rx_config_index(ctx) == 0; }

bool radix_packed_applicable(const bwts_ctx *ctx, u64 m, int key_bits)
{
    const bool packed_ok = [ctx] { const char *e = bwts_knob(ctx, "BWTS_RX_PACK"); return !(e && atoi(e) == 0); }();
    const int passes = ((key_bits < 1 ? 1 : key_bits) + 7) / 8;
    return packed_ok && rx_config_index(ctx) == 0 && passes >= 3 && passes <= 5 && m >= 65536;
}

int radix_sort_pairs(bwts_ctx *ctx, const SortPlan &plan, u64 m, int key_bits, int *result_buf)
{
    int cur = 0;
    if (m == 0) { *result_buf = 0; return BWTS_OK; }
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 64) key_bits = 64;
    const int passes = (key_bits + 7) / 8;
    const u64 tiles = radix_tiles(ctx, m);
    const int cfg = rx_config_index(ctx);
    u32 *tile_hist = plan.tile_hist;

    const bool small_ok = [ctx] { const char *e = bwts_knob(ctx, "BWTS_RX_SMALL"); return !(e && atoi(e) == 0); }();
    if (small_ok && m <= RS_MAX && !plan.sym_src && !plan.vals_identity && !plan.keys_split) {
        SpanGuard g(ctx, BWTS_K_RADIX_SCATTER, m, 24 * m);
        radix_sort_small_kernel<<<dim3(1), dim3(RS_THREADS), 0, ctx->stream>>>(plan.keys[0], plan.vals[0], plan.keys[1], plan.vals[1], (u32)m, passes);
        HIPC(hipGetLastError());
        *result_buf = 1;
        return BWTS_OK;
    }
    if (plan.keys_split) {      // round 0 of the forward transform: keybuild left the keys split for the packed passes
        if (!plan.sym_final || !plan.vals_identity || !radix_packed_applicable(ctx, m, key_bits)) return BWTS_E_INTERNAL;
        if (passes == 5) return radix_sort_packed<true>(ctx, plan, m, passes, result_buf);
        return radix_sort_packed<false>(ctx, plan, m, passes, result_buf);
    }

    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        {
            SpanGuard g(ctx, BWTS_K_RADIX_HIST, m, 8 * m);
            launch_hist(ctx, cfg, tiles, plan.keys[cur], m, shift, tile_hist);
        }
        {
            SpanGuard g(ctx, BWTS_K_RADIX_SCAN, tiles * 256, tiles * 256 * 12);
            BWTS_TRY(radix_column_scan(ctx, tile_hist, tiles, plan.scan_temp));
        }
        const bool ident = plan.vals_identity && p == 0;
        if (plan.sym_src) {
            if (cfg != 0) return BWTS_E_INTERNAL;
            const u8 *sin = p == 0 ? plan.sym_src : plan.sym_buf[(p - 1) & 1];
            u8 *sout = p == passes - 1 ? plan.sym_final : plan.sym_buf[p & 1];
            SpanGuard g(ctx, BWTS_K_RADIX_SCATTER, m, (ident ? 22 : 26) * m);
            if (ident)
                BWTS_TRY((launch_scatter2_t<512, 16, 4, true, true>(ctx, tiles, plan.keys[cur], plan.vals[cur], plan.keys[cur ^ 1],
                                                                     plan.vals[cur ^ 1], tile_hist, m, shift, sin, sout)));
            else {
                // timed a second time under its own class: the roofline kernel (one template variant, n-sized launches)
                SpanGuard gm(ctx, BWTS_K_RADIX_SCATTER_MAIN, m, 26 * m);
                BWTS_TRY((launch_scatter2_t<512, 16, 4, true, false>(ctx, tiles, plan.keys[cur], plan.vals[cur], plan.keys[cur ^ 1],
                                                                      plan.vals[cur ^ 1], tile_hist, m, shift, sin, sout)));
            }
        } else if (ident) {
            if (cfg != 0) return BWTS_E_INTERNAL;
            SpanGuard g(ctx, BWTS_K_RADIX_SCATTER, m, 20 * m);
            BWTS_TRY((launch_scatter2_t<512, 16, 4, false, true>(ctx, tiles, plan.keys[cur], plan.vals[cur], plan.keys[cur ^ 1], plan.vals[cur ^ 1],
                                                                  tile_hist, m, shift)));
        } else {
            SpanGuard g(ctx, BWTS_K_RADIX_SCATTER, m, 24 * m);
            BWTS_TRY(launch_scatter(ctx, cfg, tiles, plan.keys[cur], plan.vals[cur], plan.keys[cur ^ 1], plan.vals[cur ^ 1], tile_hist, m,
                                    shift));
        }
        HIPC(hipGetLastError());
        cur ^= 1;
    }
    *result_buf = cur;
    return BWTS_OK;
}
