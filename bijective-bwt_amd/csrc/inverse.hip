// inverse.hip -- BWTS inverse transform on the GPU.
//
// Replaces the inline core of /root/reference/unbwts.c:31-86:
//   :34-36  histogram            -> byte_hist_kernel (forward.hip)
//   :38-43  exclusive scan       -> column scan of the [tile][symbol] table (radix.hip)
//   :50-52  prev[i]=counts[B[i]]++ (stable LF map) -> lf_hist_kernel + lf_rank_kernel
//   :66-86  cycle walk, smallest unvisited index first, text written backwards
//           -> splitter walk recording every segment's symbols, reduced-list ranking by pointer jumping,
//              coalesced placement of the recorded segments
// The reference follows ONE cycle at a time (n dependent loads).  Here every G-th index is a
// splitter; a lane walks from its splitter to the next one, so ~n/G walks run concurrently, and LF is
// chased exactly once.  Cycles that contain no splitter are found from the visited marks and resolved separately.
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
#define LF_VISITED 0xffffffffu     // written over an entry once the walk has read it (LF is chased exactly once)


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
    u32 sy[LF_ITEMS];                     // loads first, LDS atomics after: 16 loads in flight per lane
#pragma unroll
    for (int j = 0; j < LF_ITEMS; j++) {
        const u64 i = base + (u64)j * LF_THREADS + tid;
        sy[j] = i < n ? (u32)B[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < LF_ITEMS; j++) {
        const u64 i = base + (u64)j * LF_THREADS + tid;
        if (i < n) atomicAdd(&bins[w][sy[j]], 1u);
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
// splitter walk (the only pass that chases LF)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ u32 symbol_of(const u64 *Ctab, u32 y)
{
    // B[x] is the symbol whose C-range holds LF[x]: largest c with Ctab[c] <= y  (Ctab[256] = n)
    u32 lo = 0, hi = 255;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const u32 mid = (lo + hi + 1) >> 1;
        if (Ctab[mid] <= (u64)y) lo = mid; else hi = mid - 1;     // 64-bit: boundaries reach n = 2^32
    }
    return lo;
}

// A lane walks LF from its splitter to the next one.  Along the way it marks the entries it visits (it
// overwrites them with LF_VISITED; a byte map is used instead when that value could be a real entry), keeps the smallest index seen, and records the symbols it passes -- B[x] read off LF[x] -- into
// its node's slot of `seg`, 4 at a time.  A walk that reaches `slot` steps without meeting a splitter closes its
// node there and continues as a fresh virtual node (ids >= s, handed out by an atomic counter), so no segment
// outgrows its slot.  A wave pulls batches of splitter ids from a shared counter and hands them to its lanes as
// they finish (one atomic per WALK_BATCH walks); every walk ends at the next splitter (LF is a permutation) and
// a lane that finds the counter exhausted stops asking, so every wave drains.
#define WALK_BATCH 128
template <bool BYTEMARK>
__global__ __launch_bounds__(256) void walk_record_kernel(u32 *__restrict__ LF, u8 *__restrict__ marks, u64 s, u64 node_cap, int g, u32 slot,
                                                          const u64 *__restrict__ Cg, u8 *__restrict__ seg,
                                                          u32 *__restrict__ nxt, u32 *__restrict__ seglen,
                                                          u32 *__restrict__ segmin, u32 *__restrict__ segminoff,
                                                          unsigned long long *__restrict__ ticket,
                                                          unsigned long long *__restrict__ vcount,
                                                          unsigned long long *__restrict__ overflow)
{
    __shared__ u64 Ctab[257];
    for (int i = threadIdx.x; i < 257; i += 256) Ctab[i] = Cg[i];
    __syncthreads();
    const u32 gmask = (1u << g) - 1u;
    bool have = false, done = false;
    u64 my = 0;
    u32 x = 0, len = 0, mn = 0, mnoff = 0;
    u32 sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0;     // 16 recorded symbols waiting for one 16-byte store
    u64 bnext = 0, bend = 0;            // the wave's current batch of splitter ids (wave-uniform)
    bool exhausted = false;
    for (;;) {
        const u64 need = __ballot(!have && !done);
        if (need) {
            if (bnext == bend && !exhausted) {
                const int leader = __ffsll((unsigned long long)need) - 1;
                unsigned long long basev = 0;
                if (lane_id() == leader) basev = atomicAdd(ticket, (unsigned long long)WALK_BATCH);
                basev = shfl_t((u64)basev, leader);
                bnext = basev;
                bend = basev + WALK_BATCH < s ? basev + WALK_BATCH : s;
                if (basev >= s) { exhausted = true; bnext = bend = 0; }
            }
            if (!have && !done) {
                const u64 id = bnext + (u64)__popcll(need & lanemask_lt());
                if (id < bend) { have = true; my = id; x = (u32)(my << g); len = 0; mn = x; mnoff = 0; sb0 = sb1 = sb2 = sb3 = 0; }
                else if (exhausted) done = true;      // no work left anywhere: this lane never asks again
            }
            const u64 taken = bnext + (u64)__popcll(need);
            bnext = taken < bend ? taken : bend;
        }
        if (__ballot(have || !done) == 0) break;     // every lane has seen the counter run dry
        if (have) {
            const u32 y = LF[x];
            if (BYTEMARK) marks[x] = 1; else LF[x] = LF_VISITED;      // the entry is not needed again; byte map only when 0xffffffff is a valid value
            {
                const u32 sh = symbol_of(Ctab, y) << (8 * (len & 3u));
                const u32 w = (len >> 2) & 3u;
                sb0 |= w == 0 ? sh : 0u; sb1 |= w == 1 ? sh : 0u; sb2 |= w == 2 ? sh : 0u; sb3 |= w == 3 ? sh : 0u;
            }
            if ((len & 15u) == 15u) {
                *(uint4 *)(seg + my * slot + (len & ~15u)) = make_uint4(sb0, sb1, sb2, sb3);
                sb0 = sb1 = sb2 = sb3 = 0;
            }
            len++;
            x = y;
            const bool at_splitter = (x & gmask) == 0;
            if (at_splitter || len == slot) {
                if (len & 15u) *(uint4 *)(seg + my * slot + (len & ~15u)) = make_uint4(sb0, sb1, sb2, sb3);   // slot is a multiple of 16
                u64 next_node;
                if (at_splitter) {
                    next_node = x >> g;
                    have = false;
                } else {
                    next_node = s + atomicAdd(vcount, 1ull);
                    if (next_node >= node_cap) { atomicAdd(overflow, 1ull); next_node = node_cap - 1; }
                }
                nxt[my] = (u32)next_node; seglen[my] = len; segmin[my] = mn; segminoff[my] = mnoff;
                if (!at_splitter) { my = next_node; len = 0; mn = x; mnoff = 0; sb0 = sb1 = sb2 = sb3 = 0; }
            } else if (x < mn) { mn = x; mnoff = len; }
        }
    }
}

// out[end_c - t] = B[LF^t(min_c)] (unbwts.c:73-82): node v's recorded symbols go to out[opos - i], wrapping to the
// cycle's end once the walk passes the cycle's smallest element.  tpn threads share a node; both sides coalesce.
__global__ __launch_bounds__(256) void place_segments_kernel(const u8 *__restrict__ seg, u64 nodes, u32 slot, int tpn_log2,
                                                             const u32 *__restrict__ seglen, const u32 *__restrict__ opos,
                                                             const u32 *__restrict__ wrap_at, const u32 *__restrict__ cyc_len,
                                                             u8 *__restrict__ out)
{
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 v = gid >> tpn_log2;
    if (v >= nodes) return;
    const u32 sub = (u32)(gid & ((1ull << tpn_log2) - 1ull)), tpn = 1u << tpn_log2;
    const u32 len = seglen[v], o = opos[v], wr = wrap_at[v], L = cyc_len[v];
    const u8 *src = seg + v * slot;
    for (u32 i = sub; i < len; i += tpn) out[i >= wr ? o - i + L : o - i] = src[i];
}

// ------------------------------------------------------------------------------------
// elements no walk reached (cycles without a splitter): one sweep, wave-aggregated append
// ------------------------------------------------------------------------------------
template <bool BYTEMARK>
__global__ __launch_bounds__(256) void collect_unvisited_kernel(const u32 *__restrict__ LF, const u8 *__restrict__ marks, u64 n,
                                                                u32 *__restrict__ uidx, u32 *__restrict__ ulf, u64 cap,
                                                                unsigned long long *__restrict__ count)
{
    for (u64 base = (u64)blockIdx.x * 256; base < n; base += (u64)gridDim.x * 256) {
        const u64 i = base + threadIdx.x;
        bool un = false;
        u32 v = 0;
        if (i < n) {
            if (BYTEMARK) { un = marks[i] == 0; if (un) v = LF[i]; }
            else { v = LF[i]; un = v != LF_VISITED; }
        }
        const u64 m = __ballot(un);
        if (m) {
            const int leader = __ffsll((unsigned long long)m) - 1;
            unsigned long long b = 0;
            if (lane_id() == leader) b = atomicAdd(count, (unsigned long long)__popcll(m));
            b = shfl_t((u64)b, leader);
            if (un) {
                const u64 at = b + (u64)__popcll(m & lanemask_lt());
                if (at < cap) { uidx[at] = (u32)i; ulf[at] = v; }
            }
        }
    }
}

__global__ __launch_bounds__(256) void scatter_bytes_kernel(const u32 *__restrict__ pos, const u8 *__restrict__ sym, u64 m, u8 *__restrict__ out)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < m) out[pos[i]] = sym[i];
}

// ------------------------------------------------------------------------------------
// reduced-list ranking (splitter nodes) by pointer jumping
// ------------------------------------------------------------------------------------
// node v: nxt[v] next splitter on its cycle, len[v] elements in its segment, mn[v] smallest element of the
// segment and off[v] its distance from the splitter.  Wanted per node: its cycle's leader (smallest node
// id), length, smallest element, and the node's distance from that smallest element along LF.
#define LR_NIL 0xffffffffu

__global__ __launch_bounds__(256) void lr_init_kernel(u64 s, const u32 *__restrict__ nxt, const u32 *__restrict__ mn,
                                                      u32 *__restrict__ lead, u32 *__restrict__ cmin, u32 *__restrict__ hop)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v < s) { lead[v] = (u32)v; cmin[v] = mn[v]; hop[v] = nxt[v]; }
}

// after r rounds a node has folded in the 2^r nodes that follow it
__global__ __launch_bounds__(256) void lr_jump_min_kernel(u64 s, const u32 *__restrict__ lead_in, const u32 *__restrict__ cmin_in,
                                                          const u32 *__restrict__ hop_in, u32 *__restrict__ lead_out,
                                                          u32 *__restrict__ cmin_out, u32 *__restrict__ hop_out)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const u32 h = hop_in[v];
    const u32 l0 = lead_in[v], l1 = lead_in[h], c0 = cmin_in[v], c1 = cmin_in[h];
    lead_out[v] = l0 < l1 ? l0 : l1;
    cmin_out[v] = c0 < c1 ? c0 : c1;
    hop_out[v] = hop_in[h];
}

// cut every cycle in front of its leader, then suffix sums of the segment lengths
__global__ __launch_bounds__(256) void lr_cut_kernel(u64 s, const u32 *__restrict__ nxt, const u32 *__restrict__ len,
                                                     const u32 *__restrict__ lead, u32 *__restrict__ sum, u32 *__restrict__ hop)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v < s) { sum[v] = len[v]; hop[v] = nxt[v] == lead[v] ? LR_NIL : nxt[v]; }
}

__global__ __launch_bounds__(256) void lr_jump_sum_kernel(u64 s, const u32 *__restrict__ sum_in, const u32 *__restrict__ hop_in,
                                                          u32 *__restrict__ sum_out, u32 *__restrict__ hop_out)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const u32 h = hop_in[v];
    if (h == LR_NIL) { sum_out[v] = sum_in[v]; hop_out[v] = LR_NIL; }
    else { sum_out[v] = sum_in[v] + sum_in[h]; hop_out[v] = hop_in[h]; }
}

struct CycleRec { u32 leader; u32 minelem; u32 len; u32 pad; };

// dist[v] = elements between the leader's splitter and v's splitter; the node whose segment holds the
// cycle's smallest element publishes that element's distance; leaders append a cycle record
__global__ __launch_bounds__(256) void lr_finish_kernel(u64 s, const u32 *__restrict__ lead, const u32 *__restrict__ cmin,
                                                        const u32 *__restrict__ sum, const u32 *__restrict__ mn,
                                                        const u32 *__restrict__ off, u32 *__restrict__ dist,
                                                        u32 *__restrict__ min_dist /* by leader */, CycleRec *__restrict__ recs,
                                                        unsigned long long *__restrict__ nrec)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const u32 l = lead[v];
    const u32 L = sum[l];
    const u32 d = L - sum[v];
    dist[v] = d;
    if (mn[v] == cmin[v]) min_dist[l] = d + off[v];
    if (l == (u32)v) {
        const unsigned long long at = atomicAdd(nrec, 1ull);
        CycleRec r; r.leader = l; r.minelem = cmin[v]; r.len = L; r.pad = 0;
        recs[at] = r;
    }
}

__global__ __launch_bounds__(256) void lr_place_kernel(u64 s, const u32 *__restrict__ lead, const u32 *__restrict__ sum,
                                                       const u32 *__restrict__ dist, const u32 *__restrict__ min_dist,
                                                       const u32 *__restrict__ end_by_leader, u32 *__restrict__ opos,
                                                       u32 *__restrict__ wrap_at, u32 *__restrict__ cyc_len)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const u32 l = lead[v];
    const u32 L = sum[l];
    const u32 dm = min_dist[l];
    const u32 d = dist[v];
    const u32 t = d >= dm ? d - dm : d + L - dm;     // distance of v's splitter from the cycle's smallest element
    opos[v] = end_by_leader[l] - t;
    wrap_at[v] = L - t;
    cyc_len[v] = L;
}

__global__ __launch_bounds__(256) void scatter_u32_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ val, u64 m, u32 *__restrict__ out)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < m) out[idx[i]] = val[i];
}

// ------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------
static int splitter_log2(u64 n)
{
    int bl = 0; for (u64 x = n; x; x >>= 1) bl++;
    int g = bl - 21;
    if (g < 4) g = 4;
    if (g > 9) g = 9;
    const char *env = getenv("BWTS_SPLIT_LOG2");
    if (env) { int v = atoi(env); if (v >= 0 && v <= 20) g = v; }
    return g;
}

#define UNV_CAP (1ull << 20)     // unvisited elements collected by the single sweep

size_t inverse_arena_bytes(u64 n)
{
    const u64 s = (n >> 4) + 2;   // upper bound on splitters (g >= 4)
    return align_up(n * 4, 256) + radix_tile_hist_bytes(n) + scan_temp_bytes(n) + 24 * align_up(s * 4, 256) + (1 << 16);
}

static int grid1(u64 m) { return (int)((m + 255) / 256); }

// One attempt with splitter spacing 2^g.  *retry is set when more elements sit in splitter-free cycles than
// the sweep collects; the caller then repeats with g = 0 (every element a splitter: plain pointer jumping).
static int inverse_attempt(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out, int g, bool bytemark, bool *retry, bool *ambiguous)
{
    *retry = false;
    *ambiguous = false;
    const u64 G = 1ull << g;
    const u64 s = (n + G - 1) / G;
    const u64 tiles = (n + LF_TILE - 1) / LF_TILE;

    // a segment longer than `slot` steps is cut into virtual nodes; room for s/8 of them (natural data needs ~2 %)
    const u32 slot = (u32)(4 * G < 16 ? 16 : 4 * G);
    const u64 node_cap = g == 0 ? s : s + s / 8 + 1024;
    const size_t s4 = align_up(node_cap * 4, 256);
    BWTS_TRY(arena_reserve(ctx, align_up(n * 4, 256) + radix_tile_hist_bytes(n) + scan_temp_bytes(n) + 24 * s4 +
                                    align_up(node_cap * sizeof(CycleRec), 256) + align_up(node_cap * slot, 256) + align_up(n, 256) +
                                    (1 << 16)));
    u32 *LF = arena_array<u32>(ctx, n);
    u32 *tile_hist = (u32 *)arena_alloc(ctx, radix_tile_hist_bytes(n));
    void *scan_temp = arena_alloc(ctx, scan_temp_bytes(n));
    u32 *node[20];
    for (int i = 0; i < 20; i++) node[i] = arena_array<u32>(ctx, node_cap);
    CycleRec *d_recs = (CycleRec *)arena_alloc(ctx, node_cap * sizeof(CycleRec));
    u8 *seg = arena_array<u8>(ctx, node_cap * slot);
    u8 *marks = bytemark ? arena_array<u8>(ctx, n) : nullptr;
    if (!LF || !tile_hist || !scan_temp || !node[19] || !d_recs || !seg || (bytemark && !marks)) return BWTS_E_NOMEM;
    if (bytemark) HIPC(hipMemsetAsync(marks, 0, n, ctx->stream));
    u32 *nxt = node[0], *seglen = node[1], *segmin = node[2], *segoff = node[3];
    u32 *d_opos = node[4], *d_wrap = node[5], *d_clen = node[6];
    u32 *lead[2] = {node[7], node[8]}, *cmin[2] = {node[9], node[10]}, *hop[2] = {node[11], node[12]};
    u32 *sum[2] = {node[13], node[14]}, *dist = node[15], *min_dist = node[16], *end_by_leader = node[17];
    u32 *tmp_idx = node[18], *tmp_val = node[19];

    // symbol boundaries C[0..256] on the host (unbwts.c:38-43)
    BWTS_TRY(byte_histogram_device(ctx, d_in, n, ctx->d_small + SMI_HIST));
    BWTS_TRY(read_small(ctx, SMI_HIST, 256));
    u64 *hC = ctx->h_small + 1024;
    {
        u64 acc = 0;
        for (int c = 0; c < 256; c++) { hC[c] = acc; acc += ctx->h_small[SMI_HIST + c]; }
        hC[256] = acc;
        if (acc != n) return BWTS_E_INTERNAL;
    }
    u64 *dC = ctx->d_small + 1024;
    HIPC(hipMemcpyAsync(dC, hC, 257 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));

    // stable LF map (unbwts.c:50-52)
    {
        SpanGuard sg(ctx, BWTS_K_LF_BUILD, n, 5 * n);
        lf_hist_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in, n, tile_hist);
        BWTS_TRY(radix_column_scan(ctx, tile_hist, tiles, scan_temp));
        lf_rank_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in, n, tile_hist, LF);
        HIPC(hipGetLastError());
    }

    // the walk: marks, segment symbols, reduced list
    unsigned long long *ticket = (unsigned long long *)(ctx->d_small + SMI_COUNTERS);
    HIPC(hipMemsetAsync(ticket, 0, 8 * sizeof(u64), ctx->stream));
    const u64 walkers = s < 524288 ? s : 524288;
    const unsigned wblocks = (unsigned)((walkers + 255) / 256);
    {
        SpanGuard sg(ctx, BWTS_K_WALK, n, 6 * n);
        if (bytemark)
            walk_record_kernel<true><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, marks, s, node_cap, g, slot, dC, seg, nxt, seglen, segmin,
                                                                                  segoff, ticket, ticket + 3, ticket + 4);
        else
            walk_record_kernel<false><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, marks, s, node_cap, g, slot, dC, seg, nxt, seglen, segmin,
                                                                                   segoff, ticket, ticket + 3, ticket + 4);
        HIPC(hipGetLastError());
    }
    // virtual nodes join the reduced list: its size is only known now
    BWTS_TRY(read_small(ctx, SMI_COUNTERS, 8));
    if (ctx->h_small[SMI_COUNTERS + 4]) { *retry = true; return BWTS_OK; }   // node pool exhausted (adversarial LF): plain pointer jumping
    const u64 s_all = s + ctx->h_small[SMI_COUNTERS + 3];

    // elements in splitter-free cycles
    char *ub = nullptr;
    BWTS_TRY(aux_reserve(ctx, 2 * align_up(UNV_CAP * 4, 256) + align_up(UNV_CAP, 256), &ub));
    u32 *uidx = (u32 *)ub, *ulf = (u32 *)(ub + align_up(UNV_CAP * 4, 256));
    {
        SpanGuard sg(ctx, BWTS_K_OTHER, n, 4 * n);
        u64 blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
        if (bytemark)
            collect_unvisited_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(LF, marks, n, uidx, ulf, UNV_CAP, ticket + 1);
        else
            collect_unvisited_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(LF, marks, n, uidx, ulf, UNV_CAP, ticket + 1);
        HIPC(hipGetLastError());
    }

    // reduced-list ranking on the device
    const int R = [&] { int b = 0; for (u64 x = s_all; x; x >>= 1) b++; return b; }();   // 2^R > s >= any cycle's node count
    int cur = 0;
    {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, s_all, 0);
        const int gb = grid1(s_all);
        lr_init_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, nxt, segmin, lead[0], cmin[0], hop[0]);
        for (int r = 0; r < R; r++, cur ^= 1)
            lr_jump_min_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, lead[cur], cmin[cur], hop[cur], lead[cur ^ 1], cmin[cur ^ 1], hop[cur ^ 1]);
        u32 *leadf = lead[cur], *cminf = cmin[cur];
        int sc = 0;
        lr_cut_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, nxt, seglen, leadf, sum[0], hop[0]);
        int hc = 0;
        u32 *hopb[2] = {hop[0], hop[1]};
        for (int r = 0; r < R; r++, sc ^= 1, hc ^= 1)
            lr_jump_sum_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, sum[sc], hopb[hc], sum[sc ^ 1], hopb[hc ^ 1]);
        lr_finish_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, leadf, cminf, sum[sc], segmin, segoff, dist, min_dist, d_recs, ticket + 2);
        HIPC(hipGetLastError());
        // keep the final buffers' identities for the placement kernel
        lead[0] = leadf; sum[0] = sum[sc];
    }
    BWTS_TRY(read_small(ctx, SMI_COUNTERS, 4));
    const u64 nu = ctx->h_small[SMI_COUNTERS + 1];
    const u64 kc = ctx->h_small[SMI_COUNTERS + 2];
    ctx->tm.unvisited = nu;
    if (kc == 0 || kc > s_all) return BWTS_E_INTERNAL;
    if (nu > UNV_CAP) { *retry = true; return BWTS_OK; }   // Theta(n) elements in tiny cycles

    // host: order the cycles by smallest element (unbwts.c:62-77); splitter-free cycles join here
    std::vector<CycleRec> recs(kc);
    HIPC(hipMemcpyAsync(recs.data(), d_recs, kc * sizeof(CycleRec), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<u32> h_uidx(nu), h_ulf(nu);
    if (nu) {
        HIPC(hipMemcpyAsync(h_uidx.data(), uidx, nu * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPC(hipMemcpyAsync(h_ulf.data(), ulf, nu * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPC(hipStreamSynchronize(ctx->stream));

    struct Cyc { u32 minelem, len; u32 leader; };   // leader = LR_NIL for a splitter-free cycle
    std::vector<Cyc> cycles;
    cycles.reserve(kc + 16);
    for (const CycleRec &r : recs) cycles.push_back(Cyc{r.minelem, r.len, r.leader});
    std::vector<u32> u_cyc(nu), u_t(nu);
    if (nu) {
        // the sweep appends in arbitrary order: sort (index, LF) pairs by index
        std::vector<u32> ord(nu);
        for (u64 q = 0; q < nu; q++) ord[q] = (u32)q;
        std::sort(ord.begin(), ord.end(), [&](u32 a, u32 b) { return h_uidx[a] < h_uidx[b]; });
        std::vector<u32> si(nu), sl(nu);
        for (u64 q = 0; q < nu; q++) { si[q] = h_uidx[ord[q]]; sl[q] = h_ulf[ord[q]]; }
        h_uidx.swap(si); h_ulf.swap(sl);
        std::vector<u8> seen(nu, 0);
        for (u64 q = 0; q < nu; q++) {
            if (seen[q]) continue;
            const u32 cid = (u32)cycles.size();
            u64 at = q;
            u32 t = 0;
            do {
                seen[at] = 1;
                u_cyc[at] = cid; u_t[at] = t++;
                const u32 nx = h_ulf[at];
                const auto it = std::lower_bound(h_uidx.begin(), h_uidx.end(), nx);
                if (it == h_uidx.end() || *it != nx) {
                    if (!bytemark && n == 0x100000000ull) { *ambiguous = true; return BWTS_OK; }   // see the length check below
                    return BWTS_E_INTERNAL;
                }
                at = (u64)(it - h_uidx.begin());
            } while (at != q);
            cycles.push_back(Cyc{h_uidx[q], t, LR_NIL});
        }
    }
    ctx->tm.factors = cycles.size();
    std::vector<u32> order(cycles.size());
    for (size_t c = 0; c < order.size(); c++) order[c] = (u32)c;
    std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return cycles[a].minelem < cycles[b].minelem; });
    std::vector<u32> cyc_end(cycles.size());
    std::vector<u32> h_lidx, h_lend;
    h_lidx.reserve(kc); h_lend.reserve(kc);
    {
        u64 used = 0;
        for (u32 c : order) {
            cyc_end[c] = (u32)(n - 1 - used);
            used += cycles[c].len;
            if (cycles[c].leader != LR_NIL) { h_lidx.push_back(cycles[c].leader); h_lend.push_back(cyc_end[c]); }
        }
        if (used != n) {
            // n = 2^32 only: the one entry whose value equals LF_VISITED sat in a splitter-free cycle and was taken for visited
            if (!bytemark && n == 0x100000000ull) { *ambiguous = true; return BWTS_OK; }
            return BWTS_E_INTERNAL;
        }
    }
    HIPC(hipMemcpyAsync(tmp_idx, h_lidx.data(), kc * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipMemcpyAsync(tmp_val, h_lend.data(), kc * 4, hipMemcpyHostToDevice, ctx->stream));
    {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, s_all, 0);
        scatter_u32_kernel<<<dim3(grid1(kc)), dim3(256), 0, ctx->stream>>>(tmp_idx, tmp_val, kc, end_by_leader);
        lr_place_kernel<<<dim3(grid1(s_all)), dim3(256), 0, ctx->stream>>>(s_all, lead[0], sum[0], dist, min_dist, end_by_leader, d_opos, d_wrap, d_clen);
        HIPC(hipGetLastError());
    }

    // the recorded segments go to their places in the text (unbwts.c:73-82)
    {
        SpanGuard sg(ctx, BWTS_K_WALK_EMIT, n, 2 * n);
        int tpn_log2 = g + 2 - 4;                       // ~16 symbols per thread at the expected segment length
        if (tpn_log2 < 0) tpn_log2 = 0;
        if (tpn_log2 > 8) tpn_log2 = 8;
        const u64 threads = s_all << tpn_log2;
        place_segments_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream>>>(seg, s_all, slot, tpn_log2, seglen,
                                                                                                      d_opos, d_wrap, d_clen, d_out);
        HIPC(hipGetLastError());
    }
    // elements of splitter-free cycles
    std::vector<u32> h_upos;
    std::vector<u8> h_usym;
    if (nu) {
        h_upos.resize(nu); h_usym.resize(nu);
        for (u64 q = 0; q < nu; q++) {
            h_upos[q] = cyc_end[u_cyc[q]] - u_t[q];
            const u64 *it = std::upper_bound(hC, hC + 257, (u64)h_ulf[q]);
            h_usym[q] = (u8)((it - hC) - 1);
        }
        u32 *d_upos = uidx;                                           // the index list is no longer needed
        u8 *d_usym = (u8 *)(ub + 2 * align_up(UNV_CAP * 4, 256));
        HIPC(hipMemcpyAsync(d_upos, h_upos.data(), nu * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPC(hipMemcpyAsync(d_usym, h_usym.data(), nu, hipMemcpyHostToDevice, ctx->stream));
        scatter_bytes_kernel<<<dim3(grid1(nu)), dim3(256), 0, ctx->stream>>>(d_upos, d_usym, nu, d_out);
        HIPC(hipGetLastError());
    }
    HIPC(hipStreamSynchronize(ctx->stream));   // host vectors above must outlive the copies
    return BWTS_OK;
}

int inverse_device_impl(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out)
{
    if (n > 0x100000000ull) return BWTS_E_RANGE;
    bool retry = false, ambiguous = false;
    bool bytemark = getenv("BWTS_BYTEMARK") != nullptr;      // test hook; otherwise only after an ambiguous first attempt
    BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, splitter_log2(n), bytemark, &retry, &ambiguous));
    if (ambiguous) {
        bytemark = true;
        BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, splitter_log2(n), bytemark, &retry, &ambiguous));
    }
    if (retry) {
        BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, 0, bytemark, &retry, &ambiguous));
        if (ambiguous) BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, 0, true, &retry, &ambiguous));
        if (retry || ambiguous) return BWTS_E_INTERNAL;
    }
    return BWTS_OK;
}
