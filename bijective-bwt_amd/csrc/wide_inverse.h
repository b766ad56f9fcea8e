// wide_inverse.h -- inverse transform of inputs beyond the 32-bit index range (n > 2^32).  Included by inverse.hip.
//
// Same plan as the main path (unbwts.c:31-86 -> stable LF map, splitter walk that records every segment's symbols, ranking
// of the reduced list, placement), with 64-bit element indices and without the main path's tuning: LF is a u64 array built
// segment by segment (the tile table's offsets are 32-bit), the unreached elements come from per-range moments (a byte map as the fallback), the reduced list
// (n / 256 nodes) is ranked by plain pointer jumping, and everything sized by node or cycle counts carries 64-bit positions.
// Memory at n = 12 GiB: LF 96 GiB + recorded segments 58 + nodes ~6 (+ marks 12 on the fallback).
#define WI_G_LOG2 8
#define WI_NIL 0xffffffffu

__device__ __forceinline__ u32 symbol_of64(const u64 *Ctab, u64 y)
{
    u32 lo = 0, hi = 255;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const u32 mid = (lo + hi + 1) >> 1;
        if (Ctab[mid] <= y) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// LF[p0 + i] for one segment: the segment's scanned tile table gives the rank inside the segment (32-bit), base64[c] turns it
// into the global one: C[c] + occurrences of c in earlier segments - first slot of c in the segment's own table
__global__ __launch_bounds__(LF_THREADS) void lf_rank_wide_kernel(const u8 *__restrict__ B, u64 count, const u32 *__restrict__ tile_off,
                                                                  const u64 *__restrict__ base64, u64 *__restrict__ LF)
{
    __shared__ u32 whist[LF_WAVES][256];
    __shared__ u64 sbase[256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u64 wave_base = (u64)blockIdx.x * LF_TILE + (u64)w * (64 * LF_ITEMS);
    for (int i = tid; i < LF_WAVES * 256; i += LF_THREADS) ((u32 *)whist)[i] = 0;
    sbase[tid] = base64[tid];
    u32 sym[LF_ITEMS], rnk[LF_ITEMS];
#pragma unroll
    for (int j = 0; j < LF_ITEMS; j++) {
        const u64 i = wave_base + (u64)j * 64 + lane;
        sym[j] = i < count ? (u32)B[i] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LF_ITEMS; j++) {
        const bool valid = wave_base + (u64)j * 64 + lane < count;
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
        if (i < count) LF[i] = sbase[sym[j]] + (u64)(whist[w][sym[j]] + rnk[j]);
    }
}

// node record: where the walk went next, how many symbols it recorded, the smallest element it saw and where
struct WiNode { u32 nxt, len; u64 mn; u32 off, pad; };

// walk_record_kernel of the main path with 64-bit elements and byte-map marks (see there for the scheme)
// MOM: no byte map; the unreached elements come from per-range moments (inverse.hip, MARK_MOMENTS) over WMOM_BUCKETS ranges kept in
// dynamic LDS (80 KB: two workgroups per CU) -- the random byte write per step was what held this walk at half the main path's rate
#define WMOM_LOG2 12
#define WMOM_BUCKETS (1u << WMOM_LOG2)
template <bool MOM>
__global__ __launch_bounds__(256) void walk_record_wide_kernel(const u64 *__restrict__ LF, u8 *__restrict__ marks, u64 s, u64 node_cap, u32 slot,
                                                               const u64 *__restrict__ Cg, u8 *__restrict__ seg, WiNode *__restrict__ nodes,
                                                               unsigned long long *__restrict__ ticket, unsigned long long *__restrict__ vcount,
                                                               unsigned long long *__restrict__ overflow, int mom_shift, unsigned long long *__restrict__ mom)
{
    __shared__ u64 Ctab[257];
    extern __shared__ __attribute__((aligned(16))) unsigned long long wmom_sm[];        // MOM: sums, sums of squares, counts
    unsigned long long *msum = wmom_sm, *msq = wmom_sm + WMOM_BUCKETS;
    u32 *mcnt = (u32 *)(wmom_sm + 2 * WMOM_BUCKETS);
    if (MOM) for (u32 b = threadIdx.x; b < WMOM_BUCKETS; b += 256) { mcnt[b] = 0; msum[b] = 0; msq[b] = 0; }
    for (int i = threadIdx.x; i < 257; i += 256) Ctab[i] = Cg[i];
    __syncthreads();
    const u64 gmask = (1ull << WI_G_LOG2) - 1ull;
    bool have = false, done = false;
    u64 my = 0, x = 0, mn = 0;
    u32 len = 0, mnoff = 0;
    u32 sb[16];                  // 64 recorded symbols, stored as one 64-byte block (see walk_record_kernel)
#pragma unroll
    for (int q = 0; q < 16; q++) sb[q] = 0;
    u64 bnext = 0, bend = 0;
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
                if (id < bend) {
                    have = true; my = id; x = my << WI_G_LOG2; len = 0; mn = x; mnoff = 0;
#pragma unroll
                    for (int q = 0; q < 16; q++) sb[q] = 0;
                }
                else if (exhausted) done = true;
            }
            const u64 taken = bnext + (u64)__popcll(need);
            bnext = taken < bend ? taken : bend;
        }
        if (__ballot(have || !done) == 0) break;
        if (have) {
            const u64 y = LF[x];
            if (MOM) {
                const u32 b = (u32)x & (WMOM_BUCKETS - 1u);                  // residue classes (see inverse.hip, MARK_MOMENTS)
                const unsigned long long o = x >> WMOM_LOG2;
                atomicAdd(&mcnt[b], 1u); atomicAdd(&msum[b], o); atomicAdd(&msq[b], o * o);
            } else marks[x] = 1;
            {
                const u32 sh = symbol_of64(Ctab, y) << (8 * (len & 3u));
                const u32 w = (len >> 2) & 15u;
#pragma unroll
                for (int q = 0; q < 16; q++) sb[q] |= w == (u32)q ? sh : 0u;
            }
            if ((len & 63u) == 63u) {
                uint4 *d = (uint4 *)(seg + my * slot + (len & ~63u));
#pragma unroll
                for (int q = 0; q < 4; q++) d[q] = make_uint4(sb[4 * q], sb[4 * q + 1], sb[4 * q + 2], sb[4 * q + 3]);
#pragma unroll
                for (int q = 0; q < 16; q++) sb[q] = 0;
            }
            len++;
            x = y;
            const bool at_splitter = (x & gmask) == 0;
            if (at_splitter || len == slot) {
                if (len & 63u) {
                    uint4 *d = (uint4 *)(seg + my * slot + (len & ~63u));
                    const u32 rem = len & 63u;
#pragma unroll
                    for (int q = 0; q < 4; q++) if ((u32)q * 16u < rem) d[q] = make_uint4(sb[4 * q], sb[4 * q + 1], sb[4 * q + 2], sb[4 * q + 3]);
                }
                u64 next_node;
                if (at_splitter) { next_node = x >> WI_G_LOG2; have = false; }
                else {
                    next_node = s + atomicAdd(vcount, 1ull);
                    if (next_node >= node_cap) { atomicAdd(overflow, 1ull); next_node = node_cap - 1; }
                }
                WiNode nd; nd.nxt = (u32)next_node; nd.len = len; nd.mn = mn; nd.off = mnoff; nd.pad = 0;
                nodes[my] = nd;
                if (!at_splitter) {
                    my = next_node; len = 0; mn = x; mnoff = 0;
#pragma unroll
                    for (int q = 0; q < 16; q++) sb[q] = 0;
                }
            } else if (x < mn) { mn = x; mnoff = len; }
        }
    }
    if (MOM) {
        __syncthreads();                  // every wave leaves the loop (the pool runs dry for all of them)
        for (u32 b = threadIdx.x; b < WMOM_BUCKETS; b += 256) {
            const u32 c = mcnt[b];
            if (c) { atomicAdd(&mom[b], (unsigned long long)c); atomicAdd(&mom[WMOM_BUCKETS + b], msum[b]); atomicAdd(&mom[2 * WMOM_BUCKETS + b], msq[b]); }
        }
    }
}

// moments_solve_kernel / moments_budget_kernel / moments_chase_kernel of the main path (inverse.hip) with 64-bit elements and WMOM_BUCKETS ranges
__global__ __launch_bounds__(1024) void moments_solve_wide_kernel(const unsigned long long *__restrict__ mom, u64 n, int shift, const u64 *__restrict__ LF,
                                                                  u64 *__restrict__ uidx, u64 *__restrict__ ulf, u64 ucap, u32 *__restrict__ def_list,
                                                                  unsigned long long *__restrict__ counters)
{
    const u64 b = (u64)blockIdx.x * 1024 + threadIdx.x;
    if (b >= WMOM_BUCKETS || b >= n) return;
    const u64 size = (n - b + WMOM_BUCKETS - 1) >> WMOM_LOG2;
    const u64 cnt = mom[b];
    if (cnt > size) { atomicAdd(&counters[11], 1ull); return; }
    const u64 d = size - cnt;
    if (d == 0) return;
    const u64 sall = size * (size - 1) / 2;
    u64 f[3] = {size - 1, size, 2 * size - 1};
    { int two = 0, three = 0; for (int i = 0; i < 3; i++) { if (!two && f[i] % 2 == 0) { f[i] /= 2; two = 1; } } for (int i = 0; i < 3; i++) { if (!three && f[i] % 3 == 0) { f[i] /= 3; three = 1; } } }
    const u64 qall = f[0] * f[1] * f[2];                                     // mod 2^64, like the sums of squares it is compared with
    const u64 A = sall - mom[WMOM_BUCKETS + b], B = qall - mom[2 * WMOM_BUCKETS + b];
    if (d == 1) {
        if (A >= size || A * A != B) { atomicAdd(&counters[11], 1ull); return; }
        const unsigned long long at = atomicAdd(&counters[1], 1ull);
        if (at < ucap) { const u64 x = (A << WMOM_LOG2) | b; uidx[at] = x; ulf[at] = LF[x]; }
    } else if (d == 2) {
        const u64 D = 2 * B - A * A;
        u64 r = (u64)sqrt((double)D);
        while (r * r > D) r--;
        while ((r + 1) * (r + 1) <= D) r++;
        const u64 o1 = (A - r) / 2, o2 = (A + r) / 2;
        if (A >= 2 * size || r * r != D || r == 0 || ((A - r) & 1) || o2 >= size || o1 * o1 + o2 * o2 != B) { atomicAdd(&counters[11], 1ull); return; }
        const unsigned long long at = atomicAdd(&counters[1], 2ull);
        if (at < ucap) { const u64 x = (o1 << WMOM_LOG2) | b; uidx[at] = x; ulf[at] = LF[x]; }
        if (at + 1 < ucap) { const u64 x = (o2 << WMOM_LOG2) | b; uidx[at + 1] = x; ulf[at + 1] = LF[x]; }
    } else {
        const unsigned long long at = atomicAdd(&counters[10], 1ull);
        def_list[at] = (u32)b;
    }
}
__global__ __launch_bounds__(256) void moments_chase_wide_kernel(const u32 *__restrict__ def_list, const unsigned long long *__restrict__ counters_in, u64 n, int shift,
                                                                 const u64 *__restrict__ LF, u32 cap, u64 *__restrict__ uidx, u64 *__restrict__ ulf, u64 ucap,
                                                                 unsigned long long *__restrict__ counters)
{
    const u64 classes = counters_in[10];
    const u64 members = (n + WMOM_BUCKETS - 1) >> WMOM_LOG2;
    const u64 per = (members + 255) / 256;
    const u64 gmask = (1ull << WI_G_LOG2) - 1ull;
    (void)shift;
    for (u64 w = blockIdx.x; w < classes * per; w += gridDim.x) {
        const u64 x0 = (((w % per) * 256 + threadIdx.x) << WMOM_LOG2) | (u64)def_list[w / per];
        bool un = false;
        if (x0 < n) {
            if ((x0 & gmask) != 0) {
                u64 y = LF[x0];
                u32 steps = 0;
                for (;;) {
                    if (y == x0) { un = true; break; }
                    if ((y & gmask) == 0) break;
                    if (++steps > cap) { atomicAdd(&counters[11], 1ull); break; }
                    y = LF[y];
                }
            }
        }
        const u64 m = __ballot(un);
        if (m) {
            const int leader = __ffsll((unsigned long long)m) - 1;
            unsigned long long bse = 0;
            if (lane_id() == leader) bse = atomicAdd(&counters[1], (unsigned long long)__popcll(m));
            bse = shfl_t((u64)bse, leader);
            if (un) { const u64 at = bse + (u64)__popcll(m & lanemask_lt()); if (at < ucap) { uidx[at] = x0; ulf[at] = LF[x0]; } }
        }
    }
}

__global__ __launch_bounds__(256) void collect_unvisited_wide_kernel(const u64 *__restrict__ LF, const u8 *__restrict__ marks, u64 n,
                                                                     u64 *__restrict__ uidx, u64 *__restrict__ ulf, u64 cap, unsigned long long *__restrict__ count)
{
    for (u64 base = (u64)blockIdx.x * 256; base < n; base += (u64)gridDim.x * 256) {
        const u64 i = base + threadIdx.x;
        const bool un = i < n && marks[i] == 0;
        const u64 m = __ballot(un);
        if (m) {
            const int leader = __ffsll((unsigned long long)m) - 1;
            unsigned long long b = 0;
            if (lane_id() == leader) b = atomicAdd(count, (unsigned long long)__popcll(m));
            b = shfl_t((u64)b, leader);
            if (un) { const u64 at = b + (u64)__popcll(m & lanemask_lt()); if (at < cap) { uidx[at] = i; ulf[at] = LF[i]; } }
        }
    }
}

// pointer jumping over the nodes: records (leader, hop, smallest element) and (sum of lengths, hop)
struct WiMin { u32 leader, hop; u64 mn; };
struct WiSum { u64 sum; u32 hop, pad; };
struct WiCycle { u64 minelem, len; u32 leader, pad; };

__global__ __launch_bounds__(256) void wi_init_kernel(u64 s, const WiNode *__restrict__ nodes, WiMin *__restrict__ rec)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v < s) { WiMin r; r.leader = (u32)v; r.hop = nodes[v].nxt; r.mn = nodes[v].mn; rec[v] = r; }
}
__global__ __launch_bounds__(256) void wi_jump_min_kernel(u64 s, const WiMin *__restrict__ in, WiMin *__restrict__ out)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const WiMin a = in[v], b = in[a.hop];
    WiMin r; r.leader = a.leader < b.leader ? a.leader : b.leader; r.mn = a.mn < b.mn ? a.mn : b.mn; r.hop = b.hop;
    out[v] = r;
}
__global__ __launch_bounds__(256) void wi_cut_kernel(u64 s, const WiNode *__restrict__ nodes, const WiMin *__restrict__ rec, WiSum *__restrict__ sh)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v < s) { WiSum r; r.sum = nodes[v].len; r.hop = nodes[v].nxt == rec[v].leader ? WI_NIL : nodes[v].nxt; r.pad = 0; sh[v] = r; }
}
__global__ __launch_bounds__(256) void wi_jump_sum_kernel(u64 s, const WiSum *__restrict__ in, WiSum *__restrict__ out)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const WiSum a = in[v];
    if (a.hop == WI_NIL) out[v] = a;
    else { const WiSum b = in[a.hop]; WiSum r; r.sum = a.sum + b.sum; r.hop = b.hop; r.pad = 0; out[v] = r; }
}
__global__ __launch_bounds__(256) void wi_finish_kernel(u64 s, const WiMin *__restrict__ rec, const WiSum *__restrict__ sh, const WiNode *__restrict__ nodes,
                                                        u64 *__restrict__ dist, u64 *__restrict__ min_dist, WiCycle *__restrict__ cyc,
                                                        unsigned long long *__restrict__ ncyc)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const WiMin r = rec[v];
    const u64 L = sh[r.leader].sum, d = L - sh[v].sum;
    dist[v] = d;
    if (nodes[v].mn == r.mn) min_dist[r.leader] = d + nodes[v].off;
    if (r.leader == (u32)v) {
        const unsigned long long at = atomicAdd(ncyc, 1ull);
        WiCycle c; c.minelem = r.mn; c.len = L; c.leader = r.leader; c.pad = 0;
        cyc[at] = c;
    }
}
// cycles without a splitter (tiny_cycle_scan_kernel of the main path)
__global__ __launch_bounds__(256) void tiny_cycle_scan_wide_kernel(const u64 *__restrict__ uidx, const u64 *__restrict__ ulf, u64 nu, const u64 *__restrict__ LF,
                                                                   u32 cap, WiCycle *__restrict__ cyc, unsigned long long *__restrict__ count,
                                                                   unsigned long long *__restrict__ overflow)
{
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    bool ismin = false;
    u64 x = 0, len = 1;
    if (q < nu) {
        x = uidx[q];
        u64 y = ulf[q];
        ismin = true;
        while (y != x) {
            if (y < x) { ismin = false; break; }
            y = LF[y];
            if (++len > cap) { atomicAdd(overflow, 1ull); ismin = false; break; }
        }
    }
    const u64 m = __ballot(ismin);
    if (m == 0) return;
    const int leader = __ffsll((unsigned long long)m) - 1;
    unsigned long long b = 0;
    if (lane_id() == leader) b = atomicAdd(count, (unsigned long long)__popcll(m));
    b = shfl_t((u64)b, leader);
    if (ismin) { WiCycle c; c.minelem = x; c.len = len; c.leader = WI_NIL; c.pad = 0; cyc[b + (u64)__popcll(m & lanemask_lt())] = c; }
}
__global__ __launch_bounds__(256) void wi_cycle_keys_kernel(const WiCycle *__restrict__ cyc, u64 m, u64 *__restrict__ keys, u32 *__restrict__ vals)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < m) { keys[i] = cyc[i].minelem; vals[i] = (u32)i; }
}
struct WiLenIn {
    const WiCycle *cyc; const u32 *order;
    __device__ __forceinline__ u64 operator()(u64 j) const { return cyc[order[j]].len; }
};
struct WiEndOut {
    const WiCycle *cyc; const u32 *order; u64 m; u64 last; u64 *end_by_leader; u64 *end_of_cyc; u64 *total;
    u64 *end_by_unit_leader;      // cycles ranked as unit nodes (leader tagged WI_UNIT_TAG), or null
    __device__ __forceinline__ void operator()(u64 j, u64 used) const
    {
        const u32 i = order[j];
        const WiCycle c = cyc[i];
        const u64 end = last - used;
        if (c.leader != WI_NIL) {
            if (c.leader & 0x80000000u) end_by_unit_leader[c.leader & 0x7fffffffu] = end;
            else end_by_leader[c.leader] = end;
        }
        end_of_cyc[i] = end;
        if (j + 1 == m) *total = used + c.len;
    }
};
// node -> text position of its first symbol, symbols until the walk passes the cycle's smallest element, cycle length
__global__ __launch_bounds__(256) void wi_place_kernel(u64 s, const WiMin *__restrict__ rec, const WiSum *__restrict__ sh, const u64 *__restrict__ dist,
                                                       const u64 *__restrict__ min_dist, const u64 *__restrict__ end_by_leader,
                                                       u64 *__restrict__ opos, u64 *__restrict__ wrap_at, u64 *__restrict__ cyc_len)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const u32 l = rec[v].leader;
    const u64 L = sh[l].sum, dm = min_dist[l], d = dist[v];
    const u64 t = d >= dm ? d - dm : d + L - dm;
    opos[v] = end_by_leader[l] - t;
    wrap_at[v] = L - t;
    cyc_len[v] = L;
}
__global__ __launch_bounds__(256) void place_segments_wide_kernel(const u8 *__restrict__ seg, u64 nodes_n, u32 slot, int tpn_log2, const WiNode *__restrict__ nodes,
                                                                  const u64 *__restrict__ opos, const u64 *__restrict__ wrap_at, const u64 *__restrict__ cyc_len,
                                                                  u8 *__restrict__ out)
{
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 v = gid >> tpn_log2;
    if (v >= nodes_n) return;
    const u32 sub = (u32)(gid & ((1ull << tpn_log2) - 1ull)), tpn = 1u << tpn_log2;
    const u32 len = nodes[v].len;
    const u64 o = opos[v], wr = wrap_at[v], L = cyc_len[v];
    const u8 *src = seg + v * slot;
    for (u32 c = sub * 16; c < len; c += tpn * 16) {
        if (c + 16 <= len && ((u64)c + 16 <= wr || (u64)c >= wr)) {
            const uint4 q = *(const uint4 *)(src + c);
            const u64 base = (u64)c >= wr ? o - c + L : o - c;
            Unaligned16 r;
            r.w[0] = __builtin_bswap32(q.w); r.w[1] = __builtin_bswap32(q.z);
            r.w[2] = __builtin_bswap32(q.y); r.w[3] = __builtin_bswap32(q.x);
            *(Unaligned16 *)(out + base - 15) = r;
        } else {
            const u32 e = c + 16 < len ? c + 16 : len;
            for (u32 i = c; i < e; i++) out[(u64)i >= wr ? o - i + L : o - i] = src[i];
        }
    }
}
__global__ __launch_bounds__(256) void tiny_place_wide_kernel(const WiCycle *__restrict__ cyc, u64 m, const u64 *__restrict__ end_of_cyc,
                                                              const u64 *__restrict__ LF, const u64 *__restrict__ Cg, u8 *__restrict__ out)
{
    __shared__ u64 Ctab[257];
    for (int i = threadIdx.x; i < 257; i += 256) Ctab[i] = Cg[i];
    __syncthreads();
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const WiCycle c = cyc[i];
    if (c.leader != WI_NIL) return;
    u64 x = c.minelem, pos = end_of_cyc[i];
    for (u64 t = 0; t < c.len; t++) {
        const u64 y = LF[x];
        out[pos--] = (u8)symbol_of64(Ctab, y);
        x = y;
    }
}

// ---- cycles without a splitter that are too long for one lane --------------------------------------------------------------
// (structured inputs: no regular splitter i * 2^WI_G_LOG2 on a cycle of tens of thousands of elements).  Every unreached
// element becomes a node of its own -- length 1, successor = the compact index of LF[x] -- and the node kernels above rank that
// list in parallel (O(nu log nu) work, about 110 bytes per unreached element, taken from the device for the call).
#define WI_UNIT_TAG 0x80000000u      // leader indices of these cycles are kept apart from the splitter nodes'
template <typename IDX>      // u64 here, u32 on the main path (inverse.hip)
__global__ __launch_bounds__(256) void wi_unit_index_kernel(const IDX *__restrict__ uidx, u64 nu, IDX *__restrict__ LF)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < nu) LF[uidx[i]] = (IDX)i;                // the entry's old value lives on in ulf[i]
}
template <typename IDX>
__global__ __launch_bounds__(256) void wi_unit_nodes_kernel(const IDX *__restrict__ uidx, const IDX *__restrict__ ulf, u64 nu, const IDX *__restrict__ LF,
                                                            WiNode *__restrict__ nodes)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < nu) { WiNode nd; nd.nxt = (u32)LF[ulf[i]]; nd.len = 1; nd.mn = uidx[i]; nd.off = 0; nd.pad = 0; nodes[i] = nd; }
}
__global__ __launch_bounds__(256) void wi_unit_tag_kernel(WiCycle *__restrict__ cyc, u64 m)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < m) cyc[i].leader |= WI_UNIT_TAG;
}
// wi_place_kernel + place_segments_wide_kernel for nodes of one symbol
template <typename IDX>
__global__ __launch_bounds__(256) void wi_unit_place_kernel(u64 nu, const WiMin *__restrict__ rec, const WiSum *__restrict__ sh, const u64 *__restrict__ dist,
                                                            const u64 *__restrict__ min_dist, const IDX *__restrict__ end_by_leader,
                                                            const IDX *__restrict__ ulf, const u64 *__restrict__ Cg, u8 *__restrict__ out)
{
    __shared__ u64 Ctab[257];
    for (int i = threadIdx.x; i < 257; i += 256) Ctab[i] = Cg[i];
    __syncthreads();
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= nu) return;
    const u32 l = rec[v].leader;
    const u64 L = sh[l].sum, dm = min_dist[l], d = dist[v];
    const u64 t = d >= dm ? d - dm : d + L - dm;
    out[(u64)end_by_leader[l] - t] = (u8)symbol_of64(Ctab, (u64)ulf[v]);
}
// device memory for one call (the rare paths): released when the call returns
struct ScopedDeviceBlock {
    bwts_ctx *ctx; char *p = nullptr;
    explicit ScopedDeviceBlock(bwts_ctx *c) : ctx(c) {}
    int take(size_t bytes)
    {
        if (hipMalloc((void **)&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return BWTS_E_NOMEM; }
        if (bytes > ctx->call_block_bytes) ctx->call_block_bytes = bytes;
        return BWTS_OK;
    }
    ~ScopedDeviceBlock() { if (p) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(p); } }
};

static int inverse_wide_attempt(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out, bool moments, bool *need_marks);
static int inverse_wide_impl(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out)
{
    // unreached elements from per-range moments; the byte map when too many are missing (or BWTS_INV_MARK=bytemap / BWTS_BYTEMARK=1)
    const char *me = bwts_knob(ctx, "BWTS_INV_MARK");
    bool moments = !((me && !strcmp(me, "bytemap")) || bwts_knob(ctx, "BWTS_BYTEMARK"));
    bool need_marks = false;
    if (moments) {
        BWTS_TRY(inverse_wide_attempt(ctx, d_in, n, d_out, true, &need_marks));
        if (!need_marks) return BWTS_OK;
    }
    return inverse_wide_attempt(ctx, d_in, n, d_out, false, &need_marks);
}
static int inverse_wide_attempt(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out, bool moments, bool *need_marks)
{
    *need_marks = false;
    if (n > (1ull << 36)) return BWTS_E_RANGE;
    const int mom_shift = WMOM_LOG2;
    const u64 G = 1ull << WI_G_LOG2;
    const u64 s = (n + G - 1) / G;
    const u32 slot = (u32)(4 * G);
    const u64 node_cap = s + s / 8 + 1024;
    if (node_cap > 0xfffffff0ull) return BWTS_E_RANGE;
    int seg_log2 = 31;
    if (const char *e = bwts_knob(ctx, "BWTS_WIDE_SEG_LOG2")) { const int v = atoi(e); if (v >= 12 && v <= 31) seg_log2 = v; }
    const u64 segn = 1ull << seg_log2;
    const u64 nseg = (n + segn - 1) / segn;
    const size_t need = align_up(n * 8, 256) + align_up(n, 256) + align_up(node_cap * slot, 256) + align_up(node_cap * sizeof(WiNode), 256) +
                        2 * align_up(node_cap * sizeof(WiMin), 256) + 2 * align_up(node_cap * sizeof(WiSum), 256) + 6 * align_up(node_cap * 8, 256) +
                        align_up(node_cap * sizeof(WiCycle), 256) + radix_tile_hist_bytes(segn) + scan_temp_bytes(segn) + (1 << 18);
    BWTS_TRY(arena_reserve(ctx, need));
    u64 *LF = arena_array<u64>(ctx, n);
    u8 *marks = moments ? (u8 *)arena_alloc(ctx, 256) : arena_array<u8>(ctx, n);
    unsigned long long *mom = (unsigned long long *)arena_array<u64>(ctx, 3 * WMOM_BUCKETS);
    u32 *def_list = arena_array<u32>(ctx, WMOM_BUCKETS);
    if (!mom || !def_list) return BWTS_E_NOMEM;
    u8 *seg = arena_array<u8>(ctx, node_cap * slot);
    WiNode *nodes = arena_array<WiNode>(ctx, node_cap);
    WiMin *wmin[2] = {arena_array<WiMin>(ctx, node_cap), arena_array<WiMin>(ctx, node_cap)};
    WiSum *wsum[2] = {arena_array<WiSum>(ctx, node_cap), arena_array<WiSum>(ctx, node_cap)};
    u64 *dist = arena_array<u64>(ctx, node_cap), *min_dist = arena_array<u64>(ctx, node_cap), *end_by_leader = arena_array<u64>(ctx, node_cap);
    u64 *opos = arena_array<u64>(ctx, node_cap), *wrap_at = arena_array<u64>(ctx, node_cap), *cyc_len = arena_array<u64>(ctx, node_cap);
    WiCycle *ncyc = arena_array<WiCycle>(ctx, node_cap);
    u32 *tile_hist = (u32 *)arena_alloc(ctx, radix_tile_hist_bytes(segn));
    void *scan_temp = arena_alloc(ctx, scan_temp_bytes(segn));
    if (!LF || !marks || !seg || !nodes || !wmin[1] || !wsum[1] || !dist || !min_dist || !end_by_leader || !opos || !wrap_at || !cyc_len || !ncyc ||
        !tile_hist || !scan_temp)
        return BWTS_E_NOMEM;

    // symbol boundaries C (unbwts.c:34-43), 64-bit
    u64 *dC = ctx->d_small + 1024, *hC = ctx->h_small + 1024;
    u64 *dBase = ctx->d_small + 1536, *hBase = ctx->h_small + 1536;
    u64 *dHist = ctx->d_small + 2048, *hHist = ctx->h_small + 2048;
    BWTS_TRY(byte_histogram_device(ctx, d_in, n, dHist));
    BWTS_TRY(read_small(ctx, 2048, 256));
    {
        u64 run = 0;
        for (int c = 0; c < 256; c++) { hC[c] = run; run += hHist[c]; }
        hC[256] = n;
        if (run != n) return BWTS_E_INTERNAL;
    }
    HIPC(hipMemcpyAsync(dC, hC, 257 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    // stable LF map (unbwts.c:50-52), segment by segment
    u64 before[256];
    for (int c = 0; c < 256; c++) before[c] = 0;
    for (u64 sg = 0; sg < nseg; sg++) {
        const u64 p0 = sg * segn, c = n - p0 < segn ? n - p0 : segn;
        const u64 tiles = (c + LF_TILE - 1) / LF_TILE;
        SpanGuard g(ctx, BWTS_K_LF_BUILD, c, 10 * c);
        BWTS_TRY(byte_histogram_device(ctx, d_in + p0, c, dHist));
        BWTS_TRY(read_small(ctx, 2048, 256));
        {
            u64 run = 0;
            for (int q = 0; q < 256; q++) { hBase[q] = hC[q] + before[q] - run; run += hHist[q]; before[q] += hHist[q]; }
        }
        HIPC(hipMemcpyAsync(dBase, hBase, 256 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
        lf_hist_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in + p0, c, tile_hist);
        BWTS_TRY(radix_column_scan(ctx, tile_hist, tiles, scan_temp));
        lf_rank_wide_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in + p0, c, tile_hist, dBase, LF + p0);
        HIPC(hipGetLastError());
        HIPC(hipStreamSynchronize(ctx->stream));            // hBase is rewritten by the next segment
    }

    unsigned long long *ticket = (unsigned long long *)(ctx->d_small + SMI_COUNTERS);
    HIPC(hipMemsetAsync(ticket, 0, 16 * sizeof(u64), ctx->stream));
    if (moments) HIPC(hipMemsetAsync(mom, 0, 3 * WMOM_BUCKETS * sizeof(u64), ctx->stream));
    else HIPC(hipMemsetAsync(marks, 0, n, ctx->stream));
    {
        SpanGuard sg(ctx, BWTS_K_WALK, n, 10 * n);
        const u64 walkers = s < 524288 ? s : 524288;
        if (moments) {
            constexpr size_t lds = (size_t)WMOM_BUCKETS * (8 + 8 + 4);
            BWTS_TRY(ensure_dyn_lds(ctx, (const void *)walk_record_wide_kernel<true>, lds));
            walk_record_wide_kernel<true><<<dim3((unsigned)((walkers + 255) / 256)), dim3(256), lds, ctx->stream>>>(LF, marks, s, node_cap, slot, dC, seg, nodes, ticket,
                                                                                                                 ticket + 3, ticket + 4, mom_shift, mom);
        } else
            walk_record_wide_kernel<false><<<dim3((unsigned)((walkers + 255) / 256)), dim3(256), 0, ctx->stream>>>(LF, marks, s, node_cap, slot, dC, seg, nodes, ticket,
                                                                                                                ticket + 3, ticket + 4, 0, nullptr);
        HIPC(hipGetLastError());
    }
    BWTS_TRY(read_small(ctx, SMI_COUNTERS, 16));
    if (ctx->h_small[SMI_COUNTERS + 4]) return BWTS_E_NOMEM;          // node pool exhausted (adversarial LF): no fallback in the wide form
    const u64 s_all = s + ctx->h_small[SMI_COUNTERS + 3];

    // elements in cycles without a splitter
    u64 ucap = ctx->unv_hint > (1ull << 20) ? ctx->unv_hint : (1ull << 20);
    if (ucap > n) ucap = n;
    u64 *uidx = nullptr, *ulf = nullptr, *end_of_cyc = nullptr;
    WiCycle *cyc = nullptr;
    auto lay_out = [&](u64 cap) -> int {
        char *ub = nullptr;
        const size_t e8 = align_up((size_t)cap * 8, 256), c24 = align_up((size_t)(node_cap + cap) * sizeof(WiCycle), 256),
                     c8 = align_up((size_t)(node_cap + cap) * 8, 256);
        BWTS_TRY(aux_reserve_slot(ctx, 0, 2 * e8 + c24 + c8, &ub));
        uidx = (u64 *)ub; ulf = (u64 *)(ub + e8);
        cyc = (WiCycle *)(ub + 2 * e8);
        end_of_cyc = (u64 *)(ub + 2 * e8 + c24);
        return BWTS_OK;
    };
    BWTS_TRY(lay_out(ucap));
    auto collect = [&]() -> int {
        SpanGuard sg(ctx, BWTS_K_OTHER, n, n);
        if (moments) {
            HIPC(hipMemsetAsync(ticket + 10, 0, 2 * sizeof(u64), ctx->stream));
            const u64 per_class = (n + WMOM_BUCKETS - 1) >> WMOM_LOG2;
            const u64 budget = (8ull << 20) > per_class ? (8ull << 20) : per_class;        // elements the search may look at (at least one class)
            moments_solve_wide_kernel<<<dim3(WMOM_BUCKETS / 1024), dim3(1024), 0, ctx->stream>>>(mom, n, mom_shift, LF, uidx, ulf, ucap, def_list, ticket);
            moments_budget_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(ticket, per_class, budget);
            moments_chase_wide_kernel<<<dim3(4096), dim3(256), 0, ctx->stream>>>(def_list, ticket, n, mom_shift, LF, 1u << 16, uidx, ulf, ucap, ticket);
        } else {
            u64 blocks = (n + 255) / 256; if (blocks > 16384) blocks = 16384;
            collect_unvisited_wide_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(LF, marks, n, uidx, ulf, ucap, ticket + 1);
        }
        HIPC(hipGetLastError());
        return BWTS_OK;
    };
    BWTS_TRY(collect());
    // pointer jumping over the nodes
    const int R = [&] { int b = 0; for (u64 x = s_all; x; x >>= 1) b++; return b; }();
    int cur = 0, sc = 0;
    {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, s_all, 0);
        const int gb = grid1(s_all);
        wi_init_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, nodes, wmin[0]);
        for (int r = 0; r < R; r++, cur ^= 1) wi_jump_min_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, wmin[cur], wmin[cur ^ 1]);
        wi_cut_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, nodes, wmin[cur], wsum[0]);
        for (int r = 0; r < R; r++, sc ^= 1) wi_jump_sum_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, wsum[sc], wsum[sc ^ 1]);
        wi_finish_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, wmin[cur], wsum[sc], nodes, dist, min_dist, ncyc, ticket + 2);
        HIPC(hipGetLastError());
    }
    BWTS_TRY(read_small(ctx, SMI_COUNTERS, 16));
    const u64 nu = ctx->h_small[SMI_COUNTERS + 1];
    const u64 kc = ctx->h_small[SMI_COUNTERS + 2];
    if (moments && ctx->h_small[SMI_COUNTERS + 11]) { *need_marks = true; return BWTS_OK; }     // too many unreached elements for the moments: the byte map
    ctx->tm.unvisited = nu;
    ctx->unv_hint = (size_t)nu;
    if (nu > n || kc == 0 || kc > s_all) return BWTS_E_INTERNAL;
    if (nu > ucap) {
        ucap = nu;
        BWTS_TRY(lay_out(ucap));
        HIPC(hipMemsetAsync(ticket + 1, 0, sizeof(u64), ctx->stream));
        BWTS_TRY(collect());
    }
    if (nu) {
        SpanGuard sg(ctx, BWTS_K_OTHER, nu, 16 * nu);
        u64 cap = (1ull << 36) / nu;
        if (cap > 64 * G) cap = 64 * G;
        if (cap < 4 * G) cap = 4 * G;
        tiny_cycle_scan_wide_kernel<<<dim3(grid1(nu)), dim3(256), 0, ctx->stream>>>(uidx, ulf, nu, LF, (u32)cap, cyc, ticket + 7, ticket + 4);
        HIPC(hipGetLastError());
    }
    BWTS_TRY(read_small(ctx, SMI_COUNTERS, 16));
    u64 kt = ctx->h_small[SMI_COUNTERS + 7];
    if (kt > nu) return BWTS_E_INTERNAL;
    // a cycle without a splitter too long for one lane: all unreached elements are ranked as nodes of one symbol instead
    const bool unit_rank = ctx->h_small[SMI_COUNTERS + 4] != 0;
    ScopedDeviceBlock ub(ctx);
    WiMin *umin[2] = {nullptr, nullptr};
    WiSum *usum[2] = {nullptr, nullptr};
    u64 *udist = nullptr, *umind = nullptr, *uend = nullptr;
    int ucur = 0, usc = 0;
    if (unit_rank) {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, nu, 0);
        if (nu >= 0x7ffffff0ull) return BWTS_E_NOMEM;
        const size_t a24 = align_up(nu * sizeof(WiNode), 256), a16 = align_up(nu * 16, 256), a8 = align_up(nu * 8, 256);
        BWTS_TRY(ub.take(a24 + 4 * a16 + 3 * a8));
        WiNode *unodes = (WiNode *)ub.p;
        umin[0] = (WiMin *)(ub.p + a24); umin[1] = (WiMin *)(ub.p + a24 + a16);
        usum[0] = (WiSum *)(ub.p + a24 + 2 * a16); usum[1] = (WiSum *)(ub.p + a24 + 3 * a16);
        udist = (u64 *)(ub.p + a24 + 4 * a16); umind = udist + a8 / 8; uend = umind + a8 / 8;
        const int gb = grid1(nu);
        const int R2 = [&] { int b = 0; for (u64 x = nu; x; x >>= 1) b++; return b; }();
        HIPC(hipMemsetAsync(ticket + 9, 0, sizeof(u64), ctx->stream));
        wi_unit_index_kernel<u64><<<dim3(gb), dim3(256), 0, ctx->stream>>>(uidx, nu, LF);
        wi_unit_nodes_kernel<u64><<<dim3(gb), dim3(256), 0, ctx->stream>>>(uidx, ulf, nu, LF, unodes);
        wi_init_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, unodes, umin[0]);
        for (int r = 0; r < R2; r++, ucur ^= 1) wi_jump_min_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, umin[ucur], umin[ucur ^ 1]);
        wi_cut_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, unodes, umin[ucur], usum[0]);
        for (int r = 0; r < R2; r++, usc ^= 1) wi_jump_sum_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, usum[usc], usum[usc ^ 1]);
        wi_finish_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, umin[ucur], usum[usc], unodes, udist, umind, cyc, ticket + 9);
        HIPC(hipGetLastError());
        BWTS_TRY(read_small(ctx, SMI_COUNTERS + 9, 1));
        kt = ctx->h_small[SMI_COUNTERS + 9];          // these cycles take the place of the one-lane scan's
        if (kt == 0 || kt > nu) return BWTS_E_INTERNAL;
        wi_unit_tag_kernel<<<dim3(grid1(kt)), dim3(256), 0, ctx->stream>>>(cyc, kt);
        HIPC(hipGetLastError());
    }
    const u64 kall = kc + kt;
    ctx->tm.factors = kall;
    {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, kall, 0);
        HIPC(hipMemcpyAsync(cyc + kt, ncyc, kc * sizeof(WiCycle), hipMemcpyDeviceToDevice, ctx->stream));
        char *sb = nullptr;
        const size_t k8 = align_up(kall * 8, 256), k4 = align_up(kall * 4, 256);
        BWTS_TRY(aux_reserve_slot(ctx, 1, 2 * k8 + 2 * k4 + radix_tile_hist_bytes(kall) + scan_temp_bytes(kall) + 4096, &sb));
        SortPlan cp;
        cp.keys[0] = (u64 *)sb; cp.keys[1] = (u64 *)(sb + k8);
        cp.vals[0] = (u32 *)(sb + 2 * k8); cp.vals[1] = (u32 *)(sb + 2 * k8 + k4);
        cp.tile_hist = (u32 *)(sb + 2 * k8 + 2 * k4);
        cp.scan_temp = sb + 2 * k8 + 2 * k4 + radix_tile_hist_bytes(kall);
        wi_cycle_keys_kernel<<<dim3(grid1(kall)), dim3(256), 0, ctx->stream>>>(cyc, kall, cp.keys[0], cp.vals[0]);
        int res = 0, kbits = 0;
        for (u64 x = n - 1; x; x >>= 1) kbits++;
        BWTS_TRY(radix_sort_pairs(ctx, cp, kall, kbits < 1 ? 1 : kbits, &res));
        WiLenIn lin{cyc, cp.vals[res]};
        WiEndOut lout{cyc, cp.vals[res], kall, n - 1, end_by_leader, end_of_cyc, ctx->d_small + SMI_COUNTERS + 13, uend};
        BWTS_TRY((device_scan<false, u64>(ctx, kall, lin, lout, OpAdd(), (u64)0, cp.scan_temp)));
        wi_place_kernel<<<dim3(grid1(s_all)), dim3(256), 0, ctx->stream>>>(s_all, wmin[cur], wsum[sc], dist, min_dist, end_by_leader, opos, wrap_at, cyc_len);
        HIPC(hipGetLastError());
    }
    {
        SpanGuard sg(ctx, BWTS_K_WALK_EMIT, n, 2 * n);
        const int tpn_log2 = WI_G_LOG2 - 4;
        const u64 threads = s_all << tpn_log2;
        place_segments_wide_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream>>>(seg, s_all, slot, tpn_log2, nodes, opos, wrap_at, cyc_len,
                                                                                                           d_out);
        if (unit_rank)
            wi_unit_place_kernel<u64><<<dim3(grid1(nu)), dim3(256), 0, ctx->stream>>>(nu, umin[ucur], usum[usc], udist, umind, uend, ulf, dC, d_out);
        else if (kt) tiny_place_wide_kernel<<<dim3(grid1(kt)), dim3(256), 0, ctx->stream>>>(cyc, kt, end_of_cyc, LF, dC, d_out);
        HIPC(hipGetLastError());
    }
    BWTS_TRY(read_small(ctx, SMI_COUNTERS + 13, 1));
    if (ctx->h_small[SMI_COUNTERS + 13] != n) return BWTS_E_INTERNAL;
    return BWTS_OK;
}
