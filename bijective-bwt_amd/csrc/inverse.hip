// inverse.hip -- BWTS inverse transform on the GPU.
//
// Replaces the inline core of /root/reference/unbwts.c:31-86:
//   :34-36  histogram            -> byte_hist_kernel (forward.hip)
//   :38-43  exclusive scan       -> column scan of the [tile][symbol] table (radix.hip)
//   :50-52  prev[i]=counts[B[i]]++ (stable LF map) -> lf_hist_kernel + lf_rank_kernel
//   :66-86  cycle walk, smallest unvisited index first, text written backwards
//           -> splitter walk (pass 1), reduced-list ranking, splitter walk (pass 2, emit)
// The reference follows ONE cycle at a time (n dependent loads).  Here every G-th index is a
// splitter; a lane walks from its splitter to the next one, so ~n/G walks run concurrently.
// Cycles that contain no splitter are found from the visited marks and resolved separately.
#include "internal.h"
#include "device_utils.h"
#include "scan_templ.h"

#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define LF_THREADS 256
#define LF_WAVES   4
#define LF_ITEMS   16
#define LF_TILE    (LF_THREADS * LF_ITEMS)
#define LF_TOP     0x80000000u
#define LF_MASK    0x7fffffffu


#define SMI_HIST     0
#define SMI_COUNTERS 320

// ------------------------------------------------------------------------------------
// stable LF map
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(LF_THREADS) void lf_hist_kernel(const u8 *__restrict__ B, u64 n, u32 *__restrict__ tile_hist)
{
    __shared__ u32 bins[LF_WAVES][256];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < LF_WAVES * 256; i += LF_THREADS) ((u32 *)bins)[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * LF_TILE;
#pragma unroll
    for (int j = 0; j < LF_ITEMS; j++) {
        const u64 i = base + (u64)j * LF_THREADS + tid;
        if (i < n) atomicAdd(&bins[w][B[i]], 1u);
    }
    __syncthreads();
    u32 s = 0;
#pragma unroll
    for (int ww = 0; ww < LF_WAVES; ww++) s += bins[ww][tid];
    tile_hist[(u64)blockIdx.x * 256 + tid] = s;
}

__global__ __launch_bounds__(LF_THREADS) void lf_rank_kernel(const u8 *__restrict__ B, u64 n, const u32 *__restrict__ tile_off,
                                                             u32 *__restrict__ LF)
{
    __shared__ u32 whist[LF_WAVES][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u64 wave_base = (u64)blockIdx.x * LF_TILE + (u64)w * (64 * LF_ITEMS);
    for (int i = tid; i < LF_WAVES * 256; i += LF_THREADS) ((u32 *)whist)[i] = 0;
    u32 sym[LF_ITEMS], rnk[LF_ITEMS];
#pragma unroll
    for (int j = 0; j < LF_ITEMS; j++) {
        const u64 i = wave_base + (u64)j * 64 + lane;
        sym[j] = i < n ? (u32)B[i] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LF_ITEMS; j++) {
        const bool valid = wave_base + (u64)j * 64 + lane < n;
        const u64 peers = match_digit8(sym[j], valid);
        const u32 before = (u32)__popcll(peers & lanemask_lt());
        const u32 cnt = (u32)__popcll(peers);
        const u32 prev = whist[w][sym[j]];
        rnk[j] = prev + before;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid && before == 0) whist[w][sym[j]] = prev + cnt;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();
    {
        u32 run = tile_off[(u64)blockIdx.x * 256 + tid];
#pragma unroll
        for (int ww = 0; ww < LF_WAVES; ww++) {
            const u32 c = whist[ww][tid];
            whist[ww][tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LF_ITEMS; j++) {
        const u64 i = wave_base + (u64)j * 64 + lane;
        if (i < n) LF[i] = whist[w][sym[j]] + rnk[j];
    }
}

// ------------------------------------------------------------------------------------
// splitter walks
// ------------------------------------------------------------------------------------
// Lanes pull splitter ids from a shared counter until none are left; every lane's walk ends
// at the next splitter (LF is a permutation), so every wave drains.
__global__ __launch_bounds__(256) void walk_mark_kernel(u32 *__restrict__ LF, u64 s, int g,
                                                        u32 *__restrict__ nxt, u32 *__restrict__ seglen,
                                                        u32 *__restrict__ segmin, u32 *__restrict__ segminoff,
                                                        unsigned long long *__restrict__ ticket)
{
    const u32 gmask = (1u << g) - 1u;
    bool have = false, done = false;
    u64 my = 0;
    u32 x = 0, len = 0, mn = 0, mnoff = 0;
    for (;;) {
        const u64 need = __ballot(!have && !done);
        if (need) {
            const int leader = __ffsll((unsigned long long)need) - 1;
            unsigned long long basev = 0;
            if (lane_id() == leader) basev = atomicAdd(ticket, (unsigned long long)__popcll(need));
            basev = shfl_t((u64)basev, leader);
            if (!have && !done) {
                my = basev + (u64)__popcll(need & lanemask_lt());
                if (my < s) { have = true; x = (u32)(my << g); len = 0; mn = x; mnoff = 0; }
                else done = true;          // work exhausted: this lane never asks again
            }
        }
        if (__ballot(have) == 0) break;
        if (have) {
            const u32 y = LF[x];
            LF[x] = y | LF_TOP;
            len++;
            x = y;
            if ((x & gmask) == 0) {
                nxt[my] = x >> g; seglen[my] = len; segmin[my] = mn; segminoff[my] = mnoff;
                have = false;
            } else if (x < mn) { mn = x; mnoff = len; }
        }
    }
}

__device__ __forceinline__ u32 symbol_of(const u32 *Ctab, u32 y)
{
    // largest c with Ctab[c] <= y  (Ctab[256] = n)
    u32 lo = 0, hi = 255;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const u32 mid = (lo + hi + 1) >> 1;
        if (Ctab[mid] <= y) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void walk_emit_kernel(const u32 *__restrict__ LF, u64 s, int g,
                                                        const u32 *__restrict__ opos, const u32 *__restrict__ wrap_at,
                                                        const u32 *__restrict__ cyc_len, const u32 *__restrict__ Cg,
                                                        u8 *__restrict__ out, unsigned long long *__restrict__ ticket)
{
    __shared__ u32 Ctab[257];
    for (int i = threadIdx.x; i < 257; i += 256) Ctab[i] = Cg[i];
    __syncthreads();
    const u32 gmask = (1u << g) - 1u;
    bool have = false, done = false;
    u64 my = 0;
    u32 x = 0, i = 0, pos = 0, wr = 0, L = 0;
    for (;;) {
        const u64 need = __ballot(!have && !done);
        if (need) {
            const int leader = __ffsll((unsigned long long)need) - 1;
            unsigned long long basev = 0;
            if (lane_id() == leader) basev = atomicAdd(ticket, (unsigned long long)__popcll(need));
            basev = shfl_t((u64)basev, leader);
            if (!have && !done) {
                my = basev + (u64)__popcll(need & lanemask_lt());
                if (my < s) { have = true; x = (u32)(my << g); i = 0; pos = opos[my]; wr = wrap_at[my]; L = cyc_len[my]; }
                else done = true;
            }
        }
        if (__ballot(have) == 0) break;
        if (have) {
            const u32 y = LF[x] & LF_MASK;
            if (i == wr) pos += L;            // passed the cycle's smallest element: wrap to the cycle's end
            out[pos] = (u8)symbol_of(Ctab, y);
            pos--; i++;
            x = y;
            if ((x & gmask) == 0) have = false;
        }
    }
}

// unvisited elements (no mark) -> compact list of indices, and their LF values
struct UnvIn { const u32 *LF; __device__ __forceinline__ u32 operator()(u64 i) const { return (LF[i] & LF_TOP) ? 0u : 1u; } };
struct UnvOut {
    const u32 *LF; u32 *uidx; u32 *ulf; u64 n; u64 cap; u64 *total;
    __device__ __forceinline__ void operator()(u64 i, u32 dst) const
    {
        const u32 v = LF[i];
        const bool un = !(v & LF_TOP);
        if (un && dst < cap) { uidx[dst] = (u32)i; ulf[dst] = v; }
        if (i + 1 == n) *total = (u64)dst + (un ? 1 : 0);
    }
};
struct UnvCountIn { const u32 *LF; __device__ __forceinline__ u64 operator()(u64 i) const { return (LF[i] & LF_TOP) ? 0ull : 1ull; } };

__global__ __launch_bounds__(256) void count_unvisited_kernel(const u32 *__restrict__ LF, u64 n, unsigned long long *__restrict__ total)
{
    u64 c = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) c += (LF[i] & LF_TOP) ? 0 : 1;
    c = wave_scan_inclusive(c, OpAdd());
    if (lane_id() == 63 && c) atomicAdd(total, (unsigned long long)c);
}

__global__ __launch_bounds__(256) void scatter_bytes_kernel(const u32 *__restrict__ pos, const u8 *__restrict__ sym, u64 m, u8 *__restrict__ out)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < m) out[pos[i]] = sym[i];
}

// ------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------
static int splitter_log2(u64 n)
{
    int bl = 0; for (u64 x = n; x; x >>= 1) bl++;
    int g = bl - 20;
    if (g < 4) g = 4;
    if (g > 10) g = 10;
    const char *env = getenv("BWTS_SPLIT_LOG2");
    if (env) { int v = atoi(env); if (v >= 1 && v <= 20) g = v; }
    return g;
}

size_t inverse_arena_bytes(u64 n)
{
    const u64 tiles = (n + LF_TILE - 1) / LF_TILE;
    const u64 s = (n >> 4) + 2;   // upper bound on splitters (g >= 4)
    return align_up(n * 4, 256) + radix_tile_hist_bytes(n) + align_up(tiles * 1024, 256) + scan_temp_bytes(n) +
           8 * align_up(s * 4, 256) + align_up(n * 8, 256) /* worst-case unvisited lists */ + (1 << 16);
}

struct CycleRec { u32 minelem; u32 len; };

int inverse_device_impl(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out)
{
    if (n >= 0x80000000ull) return BWTS_E_RANGE;
    const int g = splitter_log2(n);
    const u64 G = 1ull << g;
    const u64 s = (n + G - 1) / G;
    const u64 tiles = (n + LF_TILE - 1) / LF_TILE;

    // arena: LF, tile table, node arrays; unvisited lists are sized once their count is known
    size_t base_bytes = align_up(n * 4, 256) + radix_tile_hist_bytes(n) + scan_temp_bytes(n) + 8 * align_up(s * 4, 256) + (1 << 16);
    BWTS_TRY(arena_reserve(ctx, base_bytes));
    u32 *LF = arena_array<u32>(ctx, n);
    u32 *tile_hist = (u32 *)arena_alloc(ctx, radix_tile_hist_bytes(n));
    void *scan_temp = arena_alloc(ctx, scan_temp_bytes(n));
    u32 *nxt = arena_array<u32>(ctx, s), *seglen = arena_array<u32>(ctx, s), *segmin = arena_array<u32>(ctx, s),
        *segoff = arena_array<u32>(ctx, s);
    u32 *d_opos = arena_array<u32>(ctx, s), *d_wrap = arena_array<u32>(ctx, s), *d_clen = arena_array<u32>(ctx, s);
    if (!LF || !tile_hist || !scan_temp || !nxt || !seglen || !segmin || !segoff || !d_opos || !d_wrap || !d_clen) return BWTS_E_NOMEM;

    // symbol boundaries C[0..256] on the host (unbwts.c:38-43)
    BWTS_TRY(byte_histogram_device(ctx, d_in, n, ctx->d_small + SMI_HIST));
    BWTS_TRY(read_small(ctx, SMI_HIST, 256));
    u32 *hC = (u32 *)(ctx->h_small + 1024);
    {
        u64 sum = 0;
        for (int c = 0; c < 256; c++) { hC[c] = (u32)sum; sum += ctx->h_small[SMI_HIST + c]; }
        hC[256] = (u32)sum;
        if (sum != n) return BWTS_E_INTERNAL;
    }
    u32 *dC = (u32 *)(ctx->d_small + 1024);
    HIPC(hipMemcpyAsync(dC, hC, 257 * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));

    // stable LF map (unbwts.c:50-52)
    {
        SpanGuard sg(ctx, BWTS_K_LF_BUILD, n, 5 * n);
        lf_hist_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in, n, tile_hist);
        BWTS_TRY(radix_column_scan(ctx, tile_hist, tiles, scan_temp));
        lf_rank_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in, n, tile_hist, LF);
        HIPC(hipGetLastError());
    }

    // pass 1: walk + mark
    unsigned long long *ticket = (unsigned long long *)(ctx->d_small + SMI_COUNTERS);
    HIPC(hipMemsetAsync(ticket, 0, 4 * sizeof(u64), ctx->stream));
    u64 walkers = s < 524288 ? s : 524288;
    const unsigned wblocks = (unsigned)((walkers + 255) / 256);
    {
        SpanGuard sg(ctx, BWTS_K_WALK, n, 4 * n);
        walk_mark_kernel<<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, s, g, nxt, seglen, segmin, segoff, ticket);
        HIPC(hipGetLastError());
    }
    // unvisited count
    {
        SpanGuard sg(ctx, BWTS_K_OTHER, n, 4 * n);
        u64 blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
        count_unvisited_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(LF, n, ticket + 1);
        HIPC(hipGetLastError());
    }
    BWTS_TRY(read_small(ctx, SMI_COUNTERS, 4));
    const u64 nu = ctx->h_small[SMI_COUNTERS + 1];
    ctx->tm.unvisited = nu;

    // reduced list to the host
    std::vector<u32> h_nxt(s), h_len(s), h_min(s), h_off(s);
    HIPC(hipMemcpyAsync(h_nxt.data(), nxt, s * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(h_len.data(), seglen, s * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(h_min.data(), segmin, s * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(h_off.data(), segoff, s * 4, hipMemcpyDeviceToHost, ctx->stream));

    std::vector<u32> h_uidx, h_ulf;
    if (nu) {
        char *ub = nullptr;
        const size_t each = align_up((size_t)nu * 4, 256);
        BWTS_TRY(aux_reserve(ctx, 2 * each + align_up((size_t)nu, 256), &ub));
        u32 *uidx = (u32 *)ub, *ulf = (u32 *)(ub + each);
        {
            SpanGuard sg(ctx, BWTS_K_OTHER, n, 4 * n);
            UnvIn in{LF};
            UnvOut out{LF, uidx, ulf, n, nu, ctx->d_small + SMI_COUNTERS + 2};
            BWTS_TRY((device_scan<false, u32>(ctx, n, in, out, OpAdd(), 0u, scan_temp)));
        }
        h_uidx.resize(nu); h_ulf.resize(nu);
        HIPC(hipMemcpyAsync(h_uidx.data(), uidx, nu * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPC(hipMemcpyAsync(h_ulf.data(), ulf, nu * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPC(hipStreamSynchronize(ctx->stream));

    // ---- host: rank the reduced list (BWTS_K_LISTRANK on the host for now) --------------
    // per cycle: smallest element, length; per node: distance from the cycle's smallest element
    std::vector<CycleRec> cycles;
    std::vector<u32> node_cyc(s), node_t(s);
    {
        std::vector<u8> seen(s, 0);
        std::vector<u64> dist;   // scratch per cycle
        std::vector<u32> members;
        for (u64 v = 0; v < s; v++) {
            if (seen[v]) continue;
            members.clear(); dist.clear();
            u64 total = 0, best_d = 0;
            u32 best = 0xffffffffu;
            u64 u = v;
            do {
                seen[u] = 1;
                members.push_back((u32)u);
                dist.push_back(total);
                if (h_min[u] < best) { best = h_min[u]; best_d = total + h_off[u]; }
                total += h_len[u];
                u = h_nxt[u];
                if (u >= s) return BWTS_E_INTERNAL;
            } while (u != v);
            const u32 cid = (u32)cycles.size();
            cycles.push_back(CycleRec{best, (u32)total});
            for (size_t q = 0; q < members.size(); q++) {
                const u64 t = (dist[q] + total - best_d) % total;
                node_cyc[members[q]] = cid;
                node_t[members[q]] = (u32)t;
            }
        }
    }
    // cycles made only of unvisited elements
    std::vector<u32> u_cyc(nu), u_t(nu);
    if (nu) {
        std::vector<u8> seen(nu, 0);
        for (u64 q = 0; q < nu; q++) {
            if (seen[q]) continue;
            const u32 cid = (u32)cycles.size();
            u64 cur = q;
            u32 t = 0;
            do {
                seen[cur] = 1;
                u_cyc[cur] = cid; u_t[cur] = t++;
                const u32 nx = h_ulf[cur];
                const auto it = std::lower_bound(h_uidx.begin(), h_uidx.end(), nx);
                if (it == h_uidx.end() || *it != nx) return BWTS_E_INTERNAL;
                cur = (u64)(it - h_uidx.begin());
            } while (cur != q);
            cycles.push_back(CycleRec{h_uidx[q], t});
        }
    }
    ctx->tm.factors = cycles.size();
    // order cycles by smallest element: the first one ends the text (unbwts.c:62-77)
    std::vector<u32> order(cycles.size());
    for (size_t c = 0; c < order.size(); c++) order[c] = (u32)c;
    std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return cycles[a].minelem < cycles[b].minelem; });
    std::vector<u64> cyc_end(cycles.size());
    {
        u64 used = 0;
        for (u32 c : order) { cyc_end[c] = n - 1 - used; used += cycles[c].len; }
        if (used != n) return BWTS_E_INTERNAL;
    }
    std::vector<u32> h_opos(s), h_wrap(s), h_clen(s);
    for (u64 v = 0; v < s; v++) {
        const u32 c = node_cyc[v];
        h_opos[v] = (u32)(cyc_end[c] - node_t[v]);
        h_wrap[v] = cycles[c].len - node_t[v];
        h_clen[v] = cycles[c].len;
    }
    HIPC(hipMemcpyAsync(d_opos, h_opos.data(), s * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipMemcpyAsync(d_wrap, h_wrap.data(), s * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipMemcpyAsync(d_clen, h_clen.data(), s * 4, hipMemcpyHostToDevice, ctx->stream));

    // pass 2: walk + emit (unbwts.c:73-82)
    HIPC(hipMemsetAsync(ticket, 0, sizeof(u64), ctx->stream));
    {
        SpanGuard sg(ctx, BWTS_K_WALK_EMIT, n, 6 * n);
        walk_emit_kernel<<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, s, g, d_opos, d_wrap, d_clen, dC, d_out, ticket);
        HIPC(hipGetLastError());
    }
    // elements of splitter-free cycles
    std::vector<u32> h_upos;
    std::vector<u8> h_usym;
    if (nu) {
        h_upos.resize(nu); h_usym.resize(nu);
        for (u64 q = 0; q < nu; q++) {
            h_upos[q] = (u32)(cyc_end[u_cyc[q]] - u_t[q]);
            const u32 y = h_ulf[q];
            const u32 *it = std::upper_bound(hC, hC + 257, y);
            h_usym[q] = (u8)((it - hC) - 1);
        }
        char *ub = ctx->aux;
        const size_t each = align_up((size_t)nu * 4, 256);
        u32 *d_upos = (u32 *)ub;               // uidx no longer needed
        u8 *d_usym = (u8 *)(ub + 2 * each);
        HIPC(hipMemcpyAsync(d_upos, h_upos.data(), nu * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPC(hipMemcpyAsync(d_usym, h_usym.data(), nu, hipMemcpyHostToDevice, ctx->stream));
        scatter_bytes_kernel<<<dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_upos, d_usym, nu, d_out);
        HIPC(hipGetLastError());
    }
    HIPC(hipStreamSynchronize(ctx->stream));   // host vectors above must outlive the copies
    return BWTS_OK;
}
