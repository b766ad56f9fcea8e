// dense_rounds.h -- the rounds after round 0 when many elements are tied, TILE FORM (round 2), and what it shares with the chunk form
// (chunk_rounds.h, included right behind this file by forward.hip): group detection over a tile of the list (dg_detect), PrevSym, the
// larger-group path's scan functors, the suffix array from final ranks.  The tile form runs for lists of fewer than CH_MIN_LIST
// elements, for inputs whose chunk buffers do not fit, and as BWTS_DENSE=tiles for tests.  Part of forward.hip's translation unit.
#pragma once

// ---- later rounds with the dense rank array: group-local rounds -------------------------------------------------------
// Real text ties most positions after round 0 and keeps them tied for a dozen rounds (a position inside a repeat of
// length R leaves only when the step reaches R), while nearly all of its groups are small -- the two or three copies of a
// phrase.  A round then has to cost little per tied element.  The tied list stays group-contiguous (idx = position,
// head = first slot of the group = current rank), but its order is free, SA is not maintained (final ranks say
// everything), and one kernel does a whole round for every group of at most DG_CAP elements:
//   gather r = rank[successor] -> order the members by r (LDS, counting) -> subgroups of equal r: new head = head + number
//   of smaller members -> members that are alone are finished: their rank is final and their output byte is written;
//   the others stay, with their new head.  rank[] is only READ during a round -- a member that already shows its new rank next to a
//   group-mate that still shows the old one would order the wrong way round -- and the new ranks are applied when the round's
//   gathers are done (dg_compact_kernel here, chunk_apply_moves_kernel in the chunk form).
// Larger groups (a few per cent of the elements after the first rounds) are flagged, compacted, ordered by the radix
// sort, regrouped with a scan and put back.  A stable compaction of the surviving elements gives the next round's list.
#ifndef DG_CAP
#define DG_CAP     256         // measured 16 .. 512 (text with repeats): with the quadrupled step 256 .. 384 at 4 slots per thread are the best
#endif
#define DG_THREADS 512
#ifndef DG_ITEMS
#define DG_ITEMS   4
#endif
#define DG_SPAN    (DG_THREADS * DG_ITEMS)            // list elements a workgroup looks at
#define DG_WORDS_BACK ((DG_CAP + 63) / 64)           // 64-slot words either side of a slot's own in which its group's ends may lie
#define DG_OWN     (DG_SPAN - 2 * DG_CAP - 1)         // ... and decides: DG_CAP in front and DG_CAP + 1 behind are only looked at
#define DG_FS_LDS  1024                  // factor starts kept in LDS when there are at most this many
#define DG_CNT_BIG    8                 // counters[DG_CNT_BIG .. + DG_CNT_SPREAD): elements of larger groups, spread over many addresses
#define DG_CNT_SPREAD 1024
enum { DG_DONE = 0, DG_KEEP = 1, DG_BIG = 2, DG_MOVED = 4 };      // state: low bits = what happens to the element; DG_MOVED: its head (= rank) changed

struct PrevSym {       // T[cprev(p)] (mk_bwts_sa.c:172-188): from the P array when one was built, else through the factor list
    const u8 *P; const u8 *T; u64 n; const u32 *fstart; u64 k;
    __device__ __forceinline__ u8 operator()(u64 p) const
    {
        if (P) return P[p];
        if (!fstart) return p ? T[p - 1] : T[n - 1];
        const u64 f = factor_of(fstart, k, p);
        return fstart[f] == p ? T[factor_end(fstart, k, n, f) - 1] : T[p - 1];
    }
};

// what a thread knows about its DG_ITEMS slots after the group detection shared by dense_round_kernel and dg_minpos_kernel
struct DgSlots {
    u32 idx[DG_ITEMS], gs[DG_ITEMS], sz[DG_ITEMS], h[DG_ITEMS];
    u32 kind[DG_ITEMS];      // 0 nothing to do here, 1 member of a group handled here (slots gs .. gs + sz), 2 own element of a larger group
};
__device__ __forceinline__ void dg_detect(const u32 *__restrict__ idx, const u32 *__restrict__ head, u64 a, long long e0,
                                          u32 *hd /* LDS, DG_SPAN */, u64 *startm /* LDS, DG_SPAN / 64 */, DgSlots &ds)
{
    const int tid = threadIdx.x, lane = tid & 63;
    bool valid[DG_ITEMS];
    u32 own_idx[DG_ITEMS];
    // (head and position of every slot up front, from indices clamped into the list: a load under a condition is compiled to a branch
    // with a wait behind it, one load in flight per wave -- and the position used to be fetched at the very end, item by item)
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        const long long e = e0 + j * DG_THREADS + tid;
        valid[j] = e >= 0 && (u64)e < a;
        const u64 ec = e < 0 ? 0ull : ((u64)e < a ? (u64)e : a - 1);
        ds.h[j] = head[ec];
        own_idx[j] = idx[ec];
    }
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        if (!valid[j]) ds.h[j] = 0u;
        hd[j * DG_THREADS + tid] = ds.h[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        const u32 sl = (u32)j * DG_THREADS + tid;
        const long long e = e0 + sl;
        const bool prev_valid = e - 1 >= 0 && (u64)(e - 1) < a;
        // slot 0 has no visible predecessor: it is never decided here, its flag only has to stop nobody (not a start)
        const bool st = sl > 0 && (!valid[j] || !prev_valid || ds.h[j] != hd[sl - 1]);
        const u64 m = __ballot(st);
        if (lane == 0) startm[sl >> 6] = m;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        const u32 sl = (u32)j * DG_THREADS + tid;
        ds.kind[j] = 0; ds.idx[j] = 0; ds.gs[j] = 0; ds.sz[j] = 0;
        if (!valid[j]) continue;
        // last start at or before sl, first start after sl: bit scans over the slot's 64-slot word and DG_WORDS_BACK words either side
        const u32 w = sl >> 6, b = sl & 63u;
        const u64 cur = startm[w];
        const u64 below = b == 63 ? cur : cur & ((2ull << b) - 1ull);
        int gs = -1;
        if (below) gs = (int)(w * 64 + 63 - (u32)__clzll((long long)below));
        else {
#pragma unroll
            for (u32 d = 1; d <= DG_WORDS_BACK; d++)
                if (gs < 0 && w >= d) { const u64 pm = startm[w - d]; if (pm) gs = (int)((w - d) * 64 + 63 - (u32)__clzll((long long)pm)); }
        }
        const u64 above = b == 63 ? 0ull : cur >> (b + 1);
        int ge = -1;
        if (above) ge = (int)(sl + 1 + (u32)__ffsll((unsigned long long)above) - 1);
        else {
#pragma unroll
            for (u32 d = 1; d <= DG_WORDS_BACK; d++)
                if (ge < 0 && w + d < DG_SPAN / 64) { const u64 nm = startm[w + d]; if (nm) ge = (int)((w + d) * 64 + (u32)__ffsll((unsigned long long)nm) - 1); }
        }
        const bool small = gs >= 1 && ge >= 0 && ge - gs <= DG_CAP;
        if (small) {
            if (gs >= DG_CAP && gs < DG_CAP + DG_OWN) { ds.kind[j] = 1; ds.gs[j] = (u32)gs; ds.sz[j] = (u32)(ge - gs); }
        } else if (sl >= DG_CAP && sl < DG_CAP + DG_OWN) ds.kind[j] = 2;
        if (ds.kind[j]) ds.idx[j] = own_idx[j];
    }
}

// Locality for the rounds that follow: the list is re-ordered once so that groups come in the order of their smallest
// position.  The two copies of a repeated stretch tie position by position -- (p, q), (p + 1, q + 1), ... -- so consecutive
// groups then touch consecutive ranks (gathers) and write consecutive ranks (updates) instead of random ones.  Sort key of an
// element: its group's smallest position; larger groups go behind all of those, in their old order (n + head).
__global__ __launch_bounds__(DG_THREADS) void dg_minpos_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ head, u64 a, u64 n, int kb,
                                                               u64 *__restrict__ keys, u32 *__restrict__ vals)
{
    __shared__ u32 hd[DG_SPAN];
    __shared__ u32 pos[DG_SPAN];
    __shared__ u64 startm[DG_SPAN / 64];
    const int tid = threadIdx.x;
    const long long e0 = (long long)blockIdx.x * DG_OWN - DG_CAP;
    DgSlots ds;
    dg_detect(idx, head, a, e0, hd, startm, ds);
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++)
        if (ds.kind[j] == 1) pos[j * DG_THREADS + tid] = ds.idx[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        if (!ds.kind[j]) continue;
        const u64 e = (u64)(e0 + j * DG_THREADS + tid);
        u64 sk;
        if (ds.kind[j] == 1) {
            u32 mn = 0xffffffffu;
            for (u32 m = 0; m < ds.sz[j]; m++) { const u32 q = pos[ds.gs[j] + m]; mn = q < mn ? q : mn; }
            sk = mn;
        } else sk = n + (u64)ds.h[j];
        keys[e] = ((u64)ds.h[j] << (kb > 32 ? 32 : kb)) | (kb > 32 ? sk >> (kb - 32) : sk);      // n > 2^31: the order key drops its lowest bit
        vals[e] = ds.idx[j];
    }
}
__global__ __launch_bounds__(256) void dg_unpack_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u64 a, int kb,
                                                        u32 *__restrict__ idx, u32 *__restrict__ head)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < a) { idx[i] = vals[i]; head[i] = (u32)(keys[i] >> (kb > 32 ? 32 : kb)); }
}

// Workgroup w looks at list elements [w * DG_OWN - DG_CAP, ... + DG_SPAN): LDS slot sl <-> element e = w * DG_OWN - DG_CAP + sl,
// thread t holds slots t and t + DG_THREADS (consecutive lanes = consecutive slots, so a wave's group-start flags are one
// __ballot).  It decides the elements of slots [DG_CAP, DG_CAP + DG_OWN): a group of at most DG_CAP members that starts
// there is ordered here, whole (its members reach at most DG_CAP - 1 slots further, and the slot after them shows its end);
// an element of that range whose group is larger is flagged DG_BIG.
// counters: [2] a group split, [DG_CNT_BIG ...] elements of larger groups.  rank[] is only read here: the new ranks are applied by the
// compaction pass at the end of the round (dg_compact_kernel) -- a round's keys must all come from the same version of the
// ranks: a member that already shows its new rank next to a group-mate that still shows the old one would order the wrong way.
// NKEYS = 3: the step is quadrupled -- members are ordered by the ranks of their successors at h, 2h and 3h, all read from the
// same h-consistent rank array, so one round does the work of two doublings (half the rounds, half the list passes).
template <bool CYCLIC, int NKEYS>
__global__ __launch_bounds__(DG_THREADS) void dense_round_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ head, u64 a,
                                                                 const u32 *__restrict__ rank, u64 n, u64 h,
                                                                 const u32 *__restrict__ fstart, u64 k,
                                                                 u32 *__restrict__ oidx, u32 *__restrict__ ohead, u8 *__restrict__ state,
                                                                 PrevSym prev, u8 *__restrict__ out /* null: no emission */,
                                                                 unsigned long long *__restrict__ counters, u32 *__restrict__ tile_big)
{
    __shared__ u32 hd[DG_SPAN];              // group heads
    __shared__ u32 key[DG_SPAN];             // successor ranks of the members of the groups this workgroup orders
    __shared__ u64 key23[NKEYS == 3 ? DG_SPAN : 1];      // ... and the ranks two and three steps on
    __shared__ u64 startm[DG_SPAN / 64];     // bit = a group starts at this slot (an element outside the list counts as a start)
    __shared__ u32 fs[DG_FS_LDS];
    __shared__ u32 cnt_big, any_split;
    const int tid = threadIdx.x;
    const long long e0 = (long long)blockIdx.x * DG_OWN - DG_CAP;     // list element of slot 0
    if (tid == 0) { cnt_big = 0; any_split = 0; }
    const bool fs_lds = CYCLIC && k <= DG_FS_LDS;
    if (fs_lds) for (u32 i = tid; i < k; i += DG_THREADS) fs[i] = fstart[i];
    DgSlots ds;
    dg_detect(idx, head, a, e0, hd, startm, ds);
    u32 (&my_idx)[DG_ITEMS] = ds.idx, (&my_gs)[DG_ITEMS] = ds.gs, (&my_sz)[DG_ITEMS] = ds.sz, (&my_kind)[DG_ITEMS] = ds.kind, (&myh)[DG_ITEMS] = ds.h;
    // successor ranks of the members ordered here
    u32 my_key[DG_ITEMS];
    u64 my_key23[NKEYS == 3 ? DG_ITEMS : 1];
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        my_key[j] = 0;
        if (NKEYS == 3) my_key23[NKEYS == 3 ? j : 0] = 0;
        if (my_kind[j] != 1) continue;
        const u64 p = my_idx[j];
        if (CYCLIC) {
            u64 lo = 0, hi = k - 1;
            if (fs_lds) { while (lo < hi) { const u64 mid = (lo + hi + 1) >> 1; if ((u64)fs[mid] <= p) lo = mid; else hi = mid - 1; } }
            else lo = factor_of(fstart, k, p);
            const u64 s0 = fs_lds ? fs[lo] : fstart[lo];
            const u64 e1 = lo + 1 < k ? (u64)(fs_lds ? fs[lo + 1] : fstart[lo + 1]) : n;
            my_key[j] = rank[cyclic_successor(p, s0, e1 - s0, h)];
            if (NKEYS == 3) {
                const u32 r2 = rank[cyclic_successor(p, s0, e1 - s0, 2 * h)], r3 = rank[cyclic_successor(p, s0, e1 - s0, 3 * h)];
                my_key23[NKEYS == 3 ? j : 0] = ((u64)r2 << 32) | r3;
            }
        } else {
            const u64 q = p + h;
            my_key[j] = q < n ? rank[q] + 1u : 0u;
            if (NKEYS == 3) {
                const u64 q2 = p + 2 * h, q3 = p + 3 * h;
                const u32 r2 = q2 < n ? rank[q2] + 1u : 0u, r3 = q3 < n ? rank[q3] + 1u : 0u;
                my_key23[NKEYS == 3 ? j : 0] = ((u64)r2 << 32) | r3;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++)
        if (my_kind[j] == 1) {
            key[j * DG_THREADS + tid] = my_key[j];
            if (NKEYS == 3) key23[NKEYS == 3 ? j * DG_THREADS + tid : 0] = my_key23[NKEYS == 3 ? j : 0];
        }
    __syncthreads();
    // order inside the group by counting; results first, then the loads of the emission, then every store
    u32 dst_off[DG_ITEMS], newhead[DG_ITEMS], st_out[DG_ITEMS];
    bool alone[DG_ITEMS];
    u32 big_here = 0, split_here = 0;
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        const u32 sl = (u32)j * DG_THREADS + tid;
        dst_off[j] = sl; newhead[j] = myh[j]; st_out[j] = DG_BIG; alone[j] = false;
        if (my_kind[j] == 2) big_here++;
        if (my_kind[j] != 1) continue;
        const u32 gs = my_gs[j], sz = my_sz[j], mine = my_key[j];
        u32 less = 0, eq = 0, eq_before = 0;
        if (NKEYS == 3) {
            const u64 mine23 = my_key23[NKEYS == 3 ? j : 0];
            u32 m = 0;
            for (; m + 4 <= sz; m += 4) {              // four members' keys in flight (groups of up to DG_CAP members are ordered here)
                u32 ko[4]; u64 ko23[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { ko[q] = key[gs + m + q]; ko23[q] = key23[NKEYS == 3 ? gs + m + q : 0]; }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool same = ko[q] == mine && ko23[q] == mine23;
                    less += (ko[q] < mine || (ko[q] == mine && ko23[q] < mine23)) ? 1u : 0u;
                    eq += same ? 1u : 0u;
                    eq_before += (same && gs + m + q < sl) ? 1u : 0u;
                }
            }
            for (; m < sz; m++) {
                const u32 ko = key[gs + m];
                const u64 ko23 = key23[NKEYS == 3 ? gs + m : 0];
                const bool same = ko == mine && ko23 == mine23;
                less += (ko < mine || (ko == mine && ko23 < mine23)) ? 1u : 0u;
                eq += same ? 1u : 0u;
                eq_before += (same && gs + m < sl) ? 1u : 0u;
            }
        } else {
            for (u32 m = 0; m < sz; m++) {
                const u32 ko = key[gs + m];
                less += ko < mine ? 1u : 0u;
                eq += ko == mine ? 1u : 0u;
                eq_before += (ko == mine && gs + m < sl) ? 1u : 0u;
            }
        }
        dst_off[j] = gs + less + eq_before;
        newhead[j] = myh[j] + less;
        alone[j] = eq == 1;
        st_out[j] = (alone[j] ? DG_DONE : DG_KEEP) | (less ? DG_MOVED : 0);
        split_here |= eq < sz ? 1u : 0u;
    }
    u32 pv[DG_ITEMS];
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) pv[j] = (out && alone[j]) ? (u32)prev(my_idx[j]) : 0u;
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        if (!my_kind[j]) continue;
        const u64 dst = (u64)(e0 + (long long)dst_off[j]);
        oidx[dst] = my_idx[j]; ohead[dst] = newhead[j]; state[dst] = (u8)st_out[j];
        if (out && alone[j]) out[newhead[j]] = (u8)pv[j];
    }
    if (big_here) atomicAdd(&cnt_big, big_here);
    if (split_here) any_split = 1;
    __syncthreads();
    if (tid == 0) {
        tile_big[blockIdx.x] = cnt_big;          // where the larger groups' elements are: their path then costs by their number, not by the list's
        // (one shared counter cost ~75 ns per workgroup: device-scope atomics on one address serialise across the XCDs)
        if (cnt_big) atomicAdd(&counters[DG_CNT_BIG + (blockIdx.x & (DG_CNT_SPREAD - 1))], (unsigned long long)cnt_big);
        if (any_split && __hip_atomic_load(&counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            __hip_atomic_store(&counters[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// larger groups.  dense_round_kernel left, per workgroup, how many elements of its range it flagged (tile_big, scanned into
// offsets): dg_big_collect_kernel lists the flagged elements' list indices in order, touching only ranges that hold any, and
// the rest of the path runs over that list -- group ordinals by a scan, successor-rank gather, (ordinal, rank) keys.
__global__ __launch_bounds__(256) void dg_big_collect_kernel(const u8 *__restrict__ state, u64 a, const u32 *__restrict__ tile_off, u32 *__restrict__ bigidx)
{
    __shared__ u32 scan_sm[4];
    const u32 lo = tile_off[blockIdx.x], cnt = tile_off[blockIdx.x + 1] - lo;
    if (cnt == 0) return;
    const u64 e0 = (u64)blockIdx.x * DG_OWN;
    constexpr int PER = (DG_OWN + 255) / 256;                         // blocked: PER consecutive elements per thread
    u32 f[PER], mine = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const u32 r = threadIdx.x * PER + j;
        const u64 e = e0 + r;
        f[j] = (r < DG_OWN && e < a && (state[e] & 3) == DG_BIG) ? 1u : 0u;
        mine += f[j];
    }
    u32 total;
    u32 at = lo + block_scan_exclusive<u32, OpAdd, 4>(mine, OpAdd(), 0u, scan_sm, &total);
#pragma unroll
    for (int j = 0; j < PER; j++) if (f[j]) bigidx[at++] = (u32)(e0 + threadIdx.x * PER + j);
}
struct DgBigIn {
    const u32 *bigidx; const u32 *head;
    __device__ __forceinline__ u32 operator()(u64 j) const { return (j == 0 || head[bigidx[j]] != head[bigidx[j - 1]]) ? 1u : 0u; }
};
template <bool CYCLIC>
struct DgBigOut {
    const u32 *bigidx; const u32 *head; const u32 *idx; int rb; const u32 *rank; u64 n; u64 h; const u32 *fstart; u64 k;
    u64 *bk; u32 *bv;
    u64 *k23;        // quadrupled step: (rank two steps on) << rb | (rank three steps on), kept for the regrouping ...
    u64 *k23_sort;   // ... and a copy that the first of the two sorts consumes, with the elements' indices beside it
    u32 *j_sort;
    __device__ __forceinline__ void operator()(u64 j, u32 before) const
    {
        const u32 i = bigidx[j];
        const u32 st = (j == 0 || head[i] != head[bigidx[j - 1]]) ? 1u : 0u;
        const u64 ord = (u64)before + st - 1;
        const u64 p = idx[i];
        u64 r1, r2 = 0, r3 = 0;
        if (CYCLIC) {
            const u64 f = factor_of(fstart, k, p);
            const u64 s0 = fstart[f], L = factor_end(fstart, k, n, f) - s0;
            r1 = rank[cyclic_successor(p, s0, L, h)];
            if (k23) { r2 = rank[cyclic_successor(p, s0, L, 2 * h)]; r3 = rank[cyclic_successor(p, s0, L, 3 * h)]; }
        } else {
            const u64 q = p + h;
            r1 = q < n ? (u64)rank[q] + 1ull : 0ull;
            if (k23) {
                const u64 q2 = p + 2 * h, q3 = p + 3 * h;
                r2 = q2 < n ? (u64)rank[q2] + 1ull : 0ull;
                r3 = q3 < n ? (u64)rank[q3] + 1ull : 0ull;
            }
        }
        bk[j] = (ord << rb) | r1;
        bv[j] = (u32)p;
        if (k23) { const u64 v = (r2 << rb) | r3; k23[j] = v; k23_sort[j] = v; j_sort[j] = (u32)j; }
    }
};
// between the two sorts of the quadrupled step: the elements, ordered by their second key, take their first key along
__global__ __launch_bounds__(256) void dg_stage2_keys_kernel(const u32 *__restrict__ jsorted, const u64 *__restrict__ bk, u64 m, u64 *__restrict__ keys2)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < m) keys2[i] = bk[jsorted[i]];
}
// regrouping of the sorted larger groups.  Scan value (max on both halves): high word = 1 + index of the element's group
// start, low word = 1 + index of its subgroup start, both in the sorted compacted order.
struct OpMax2 {
    template <typename T> __device__ __forceinline__ T operator()(T x, T y) const
    {
        const u32 xh = (u32)(x >> 32), yh = (u32)(y >> 32), xl = (u32)x, yl = (u32)y;
        return ((u64)(xh > yh ? xh : yh) << 32) | (u64)(xl > yl ? xl : yl);
    }
};
// (quadrupled step: bk = the sorted first keys, src[j] = the element that landed in slot j, k23 = second keys by element, pos = positions by
// element; single step: src and k23 are null and bv holds the positions in sorted order)
struct DgRegroupIn {
    const u64 *bk; u64 m; int rb; const u32 *src; const u64 *k23;
    __device__ __forceinline__ bool same(u64 a, u64 b) const { return bk[a] == bk[b] && (!k23 || k23[src[a]] == k23[src[b]]); }
    __device__ __forceinline__ u64 operator()(u64 j) const
    {
        const bool gstart = j == 0 || (bk[j - 1] >> rb) != (bk[j] >> rb);
        const bool sstart = gstart || !same(j - 1, j);
        return ((u64)(gstart ? (u32)j + 1u : 0u) << 32) | (u64)(sstart ? (u32)j + 1u : 0u);
    }
};
struct DgRegroupOut {
    const u64 *bk; const u32 *bv; const u32 *bpos; u64 m; int rb;
    u32 *oidx; u32 *ohead; u8 *state; PrevSym prev; u8 *out; unsigned long long *counters;
    const u32 *src; const u64 *k23;
    __device__ __forceinline__ void operator()(u64 j, u64 v) const       // inclusive scan value
    {
        const u32 gidx = (u32)(v >> 32) - 1u, sidx = (u32)v - 1u;
        const bool last_of_sub = j + 1 == m || bk[j + 1] != bk[j] || (k23 && k23[src[j + 1]] != k23[src[j]]);
        const bool alone = sidx == (u32)j && last_of_sub;
        // the j-th flagged list slot: sorting keeps every group on its own slots, and all of them still hold the group's old head
        const u32 at = bpos[j];
        const u32 newhead = ohead[at] + (sidx - gidx);
        const u32 p = src ? bv[src[j]] : bv[j];
        oidx[at] = p; ohead[at] = newhead; state[at] = (u8)((alone ? DG_DONE : DG_KEEP) | (sidx != gidx ? DG_MOVED : 0));
        if (alone && out) out[newhead] = prev(p);
        const u64 splitm = __ballot(sidx != gidx);
        if (splitm && lane_id() == __ffsll((unsigned long long)__ballot(true)) - 1 &&
            __hip_atomic_load(&counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            __hip_atomic_store(&counters[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// end of a round: the new ranks are written, the elements that stay are compacted (stable) into the next round's list.
// Two sweeps: per-tile counts of the elements that stay (the state bytes only), exclusive scan of the counts, then the
// move -- every load of a thread before its first store (the generic scan's output functor runs element by element, and
// its stores kept the next element's loads waiting).
#define DC_THREADS 256
#define DC_ITEMS   8
#define DC_TILE    (DC_THREADS * DC_ITEMS)
__global__ __launch_bounds__(DC_THREADS) void dg_count_kernel(const u8 *__restrict__ state, u64 a, u32 *__restrict__ partial)
{
    __shared__ u32 wsum[DC_THREADS / 64];
    const u64 base = (u64)blockIdx.x * DC_TILE;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < DC_ITEMS; j++) {
        const u64 i = base + (u64)j * DC_THREADS + threadIdx.x;
        if (i < a && (state[i] & 3) == DG_KEEP) c++;
    }
    c = wave_scan_inclusive(c, OpAdd());
    if (lane_id() == 63) wsum[wave_id()] = c;
    __syncthreads();
    if (threadIdx.x == 0) { u32 t = 0; for (int w = 0; w < DC_THREADS / 64; w++) t += wsum[w]; partial[blockIdx.x] = t; }
}
__global__ __launch_bounds__(DC_THREADS) void dg_compact_kernel(const u8 *__restrict__ state, const u32 *__restrict__ oidx,
                                                                const u32 *__restrict__ ohead, u64 a, const u32 *__restrict__ partial,
                                                                u32 *__restrict__ rank, u32 *__restrict__ n_idx, u32 *__restrict__ n_head,
                                                                u64 *__restrict__ count)
{
    __shared__ u32 scan_sm[DC_THREADS / 64];
    // blocked: thread t owns elements [8 t, 8 t + 8) of the tile, so its kept elements are consecutive in the output
    const u64 i0 = (u64)blockIdx.x * DC_TILE + (u64)threadIdx.x * DC_ITEMS;
    u32 st[DC_ITEMS], p[DC_ITEMS], hd[DC_ITEMS];
    if (i0 + DC_ITEMS <= a) {
        const uint2 sv = *(const uint2 *)(state + i0);                      // i0 is a multiple of 8
        const uint4 p0 = *(const uint4 *)(oidx + i0), p1 = *(const uint4 *)(oidx + i0 + 4);
        const uint4 h0 = *(const uint4 *)(ohead + i0), h1 = *(const uint4 *)(ohead + i0 + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) { st[j] = (sv.x >> (8 * j)) & 255u; st[4 + j] = (sv.y >> (8 * j)) & 255u; }
        p[0] = p0.x; p[1] = p0.y; p[2] = p0.z; p[3] = p0.w; p[4] = p1.x; p[5] = p1.y; p[6] = p1.z; p[7] = p1.w;
        hd[0] = h0.x; hd[1] = h0.y; hd[2] = h0.z; hd[3] = h0.w; hd[4] = h1.x; hd[5] = h1.y; hd[6] = h1.z; hd[7] = h1.w;
    } else {
#pragma unroll
        for (int j = 0; j < DC_ITEMS; j++) {
            const u64 i = i0 + j;
            st[j] = i < a ? state[i] : (u32)DG_DONE;
            p[j] = i < a ? oidx[i] : 0u;
            hd[j] = i < a ? ohead[i] : 0u;
        }
    }
    u32 mine = 0;
#pragma unroll
    for (int j = 0; j < DC_ITEMS; j++) mine += (st[j] & 3) == DG_KEEP ? 1u : 0u;
    u32 total;
    u32 at = partial[blockIdx.x] + block_scan_exclusive<u32, OpAdd, DC_THREADS / 64>(mine, OpAdd(), 0u, scan_sm, &total);
#pragma unroll
    for (int j = 0; j < DC_ITEMS; j++) {
        if (st[j] & DG_MOVED) rank[p[j]] = hd[j];
        if ((st[j] & 3) == DG_KEEP) { n_idx[at] = p[j]; n_head[at] = hd[j]; at++; }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == DC_THREADS - 1) *count = (u64)at;
}

// what is left when no group splits any more: equal infinite words (mk_bwts_sa.c ties only between identical rotations, which
// emit identical bytes).  The members of such a group take the group's slots in list order.
struct DgRestIn {
    const u32 *head;
    __device__ __forceinline__ u32 operator()(u64 i) const { return (i == 0 || head[i] != head[i - 1]) ? (u32)i + 1u : 0u; }
};
struct DgRestOut {
    const u32 *idx; const u32 *head; PrevSym prev; u8 *out; u32 *SA;
    __device__ __forceinline__ void operator()(u64 i, u32 v) const       // inclusive max-scan: 1 + list index of the group's first element
    {
        const u32 slot = head[i] + ((u32)i - (v - 1u));
        if (out) out[slot] = prev(idx[i]);
        if (SA) SA[slot] = idx[i];
    }
};
__global__ __launch_bounds__(256) void sa_from_rank_kernel(const u32 *__restrict__ rank, u64 n, u32 *__restrict__ SA)
{
    for (u64 p = (u64)blockIdx.x * 256 + threadIdx.x; p < n; p += (u64)gridDim.x * 256) SA[rank[p]] = (u32)p;
}

struct ActiveList { u32 *idx, *slot, *head; };

static int build_ranks(bwts_ctx *ctx, const u32 *SA, u64 n, const ActiveList &l, u64 a, u32 *rank)
{
    SpanGuard g(ctx, BWTS_K_RERANK, n, 8 * n);
    u64 blocks = (n + 255) / 256; if (blocks > 16384) blocks = 16384;
    rank_from_sa_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(SA, n, rank);
    if (a) rank_from_list_kernel<<<dim3((unsigned)((a + 255) / 256)), dim3(256), 0, ctx->stream>>>(l.idx, l.head, a, rank);
    HIPC(hipGetLastError());
    return BWTS_OK;
}

// ---- which round does a group have to take part in? ------------------------------------------------------------------------
// With the list ordered by position, the copies of a repeated stretch appear as RUNS: consecutive groups G_j, G_j+1, ... whose
// members are those of the previous group shifted by one position (and no member is a factor's first position).  In a round with
// step h the successors of G_j's members are then exactly the members of G_j+h -- which, as long as that group has not been
// worked on, are tied with each other: G_j cannot split.  Only the last h groups of what is left of a run can.  Rounds run with
// steps h0, 2 h0, 4 h0, ...: the group D groups away from its run's end (D = 1 for the last one) is first able to split in the
// round r with h0 (2^r - 1) < D <= h0 (2^(r+1) - 1), a function of D alone.  So every group gets its ACTIVATION ROUND once, the
// list is partitioned by it (stable), and round r works on what earlier rounds left tied plus the groups activated in r -- a
// position inside a repeat of length R is touched when the step reaches it, not in each of the log R rounds before.
// (Nothing here is a heuristic: a group that is not yet active provably does not split in that round.)
#define DG_MAX_ACT 48
template <bool CYCLIC>
__global__ __launch_bounds__(DG_THREADS) void dg_runflags_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ head, u64 a, u64 n,
                                                                 const u32 *__restrict__ fstart, u64 k, u8 *__restrict__ gflag /* bit 0: a group starts here; bit 1: a run starts here */)
{
    __shared__ u32 hd[DG_SPAN];
    __shared__ u32 pos[DG_SPAN];
    __shared__ u64 startm[DG_SPAN / 64];
    const int tid = threadIdx.x;
    const long long e0 = (long long)blockIdx.x * DG_OWN - DG_CAP;
    DgSlots ds;
    dg_detect(idx, head, a, e0, hd, startm, ds);
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        const long long e = e0 + j * DG_THREADS + tid;
        pos[j * DG_THREADS + tid] = (e >= 0 && (u64)e < a) ? idx[e] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        const u32 sl = (u32)j * DG_THREADS + tid;
        const long long e = e0 + sl;
        if (sl < DG_CAP || sl >= DG_CAP + DG_OWN || e < 0 || (u64)e >= a) continue;
        const bool gstart = (startm[sl >> 6] >> (sl & 63u)) & 1ull;
        u32 g = 0;
        if (gstart) {
            g = 3;                                               // a run starts here unless the group continues the one before it
            if (ds.kind[j] == 1 && ds.gs[j] == sl && sl >= ds.sz[j] + 1) {
                const u32 sz = ds.sz[j], ps = sl - sz;           // the group before this one must be exactly slots [ps, sl)
                bool cont = (e0 + (long long)ps >= 0) && ((startm[ps >> 6] >> (ps & 63u)) & 1ull);
                for (u32 m = 1; m < sz && cont; m++) cont = !((startm[(ps + m) >> 6] >> ((ps + m) & 63u)) & 1ull);
                for (u32 m = 0; m < sz && cont; m++) {
                    const u32 q = pos[sl + m];
                    cont = pos[ps + m] + 1u == q && q != 0u;
                    if (CYCLIC && cont) { const u64 f = factor_of(fstart, k, q); cont = fstart[f] != q; }
                }
                if (cont) g = 1;
            }
        }
        gflag[e] = (u8)g;
    }
}
struct DgRunIn {
    const u8 *gflag;
    __device__ __forceinline__ u64 operator()(u64 e) const { const u32 g = gflag[e]; return ((u64)((g >> 1) & 1u) << 32) | (u64)(g & 1u); }
};
struct DgRunOut {
    const u8 *gflag; u32 *gord; u32 *rid; u32 *gfirst; u64 a; u64 *totals;
    __device__ __forceinline__ void operator()(u64 e, u64 v) const            // inclusive: groups and runs started up to here
    {
        const u32 go = (u32)v - 1u, ri = (u32)(v >> 32) - 1u;
        gord[e] = go; rid[e] = ri;
        if (gflag[e] & 2u) gfirst[ri] = go;
        if (e + 1 == a) { totals[0] = (u64)(u32)v; totals[1] = v >> 32; }
    }
};
__global__ __launch_bounds__(256) void dg_actkeys_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ head, const u32 *__restrict__ gord,
                                                         const u32 *__restrict__ rid, const u32 *__restrict__ gfirst, u64 a, const u64 *__restrict__ totals,
                                                         u64 h0, u64 *__restrict__ keys, u32 *__restrict__ vals)
{
    const u64 e = (u64)blockIdx.x * 256 + threadIdx.x;
    if (e >= a) return;
    const u64 ngroups = totals[0], nruns = totals[1];
    const u32 ri = rid[e];
    const u64 glast = ((u64)ri + 1 < nruns ? (u64)gfirst[ri + 1] : ngroups) - 1;
    const u64 D = glast - (u64)gord[e] + 1;                      // groups from this one to the end of its run, itself included
    u32 act = 0;
    for (u64 lim = h0; D > lim && act < DG_MAX_ACT - 1; lim = 2 * lim + h0) act++;
    keys[e] = ((u64)head[e] << 8) | act;
    vals[e] = idx[e];
}
__global__ __launch_bounds__(256) void dg_unpack8_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u64 a,
                                                         u32 *__restrict__ idx, u32 *__restrict__ head)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < a) { idx[i] = vals[i]; head[i] = (u32)(keys[i] >> 8); }
}

// The rounds after round 0 when many elements are tied (dense rank array in sp.rank): see dense_round_kernel.
// cur: the tied list left by round 0 (group-contiguous).  On return the ranks in sp.rank are final (members of a group
// of equal infinite words share their group's first slot); with need_sa the suffix array is rebuilt from them.
template <bool CYCLIC>
static int dense_rounds(bwts_ctx *ctx, const u8 *d_T, u64 n, const Alphabet &al, const u32 *d_fstart, u64 k, SortSpace &sp,
                        ActiveList cur, u64 a, u32 *SA, bool need_sa, u32 *rounds_io)
{
    if (a > 0xffffffffull) return BWTS_E_NOMEM;     // every position tied at n = 2^32: beyond what the side buffers hold
    u64 *cnt = ctx->d_small + SM_DGCNT;
    const size_t e4 = align_up((size_t)a * 4, 256), e1 = align_up((size_t)a, 256);
    char *base = nullptr;
    const size_t tb4 = align_up(((size_t)a / DG_OWN + 3) * 4, 256);
    BWTS_TRY(aux_reserve(ctx, 8 * e4 + e1 + tb4, &base));
    u32 *tile_big = (u32 *)(base + 8 * e4 + e1);
    u32 *t_idx = (u32 *)base, *t_head = (u32 *)(base + e4);
    u8 *state = (u8 *)(base + 2 * e4);
    ActiveList sets[3];        // [0]: the list after round 0 in its final order (the master); [1], [2]: the rounds' working lists
    for (int i = 0; i < 3; i++) {
        sets[i].idx = (u32 *)(base + 2 * e4 + e1 + (size_t)(2 * i) * e4);
        sets[i].head = (u32 *)(base + 2 * e4 + e1 + (size_t)(2 * i + 1) * e4);
        sets[i].slot = nullptr;
    }
    const int rb = CYCLIC ? bitlen_u64(n - 1) : bitlen_u64(n);
    PrevSym prev{sp.carry_src, d_T, n, d_fstart, k};
    u8 *out = CYCLIC ? sp.carry_out : nullptr;
    u32 rounds = *rounds_io;
    int nxt = 0;
    const u64 a0 = a;
    bool by_rounds = false;                  // the list is partitioned by activation round (master), cur holds what is active
    ActiveList master{nullptr, nullptr, nullptr};
    u64 act_start[DG_MAX_ACT + 1];
    for (int r = 0; r <= DG_MAX_ACT; r++) act_start[r] = 0;
    const int kb0 = bitlen_u64(2 * n - 1), kb = kb0 > 32 ? kb0 : (kb0 + 7) / 8 * 8;      // (a whole number of digits: the sort then orders by the position field alone)
    if (a >= (1ull << 16)) {
        // groups in the order of their smallest position (see dg_minpos_kernel); the sorted list lands in sets[0]
        char *ob = nullptr;
        const size_t a8 = align_up((size_t)a * 8, 256);
        const int orc = aux_reserve_slot(ctx, 1, 2 * a8 + 2 * e4, &ob);
        if (orc != BWTS_OK && orc != BWTS_E_NOMEM) return orc;
        if (orc == BWTS_OK) {      // (no room for the sort buffers, e.g. text at n = 2^32: the rounds run on the list as it is)
            SortPlan op;
            op.keys[0] = (u64 *)ob; op.keys[1] = (u64 *)(ob + a8);
            op.vals[0] = (u32 *)(ob + 2 * a8); op.vals[1] = (u32 *)(ob + 2 * a8 + e4);
            op.tile_hist = sp.tile_hist; op.scan_temp = sp.scan_temp;
            {
                SpanGuard g(ctx, BWTS_K_RERANK, a, 20 * a);
                dg_minpos_kernel<<<dim3((unsigned)((a + DG_OWN - 1) / DG_OWN)), dim3(DG_THREADS), 0, ctx->stream>>>(cur.idx, cur.head, a, n, kb, op.keys[0], op.vals[0]);
                HIPC(hipGetLastError());
            }
            // (ordering by fewer bits was measured: 16 bits cost 30 ms more in the rounds than the two saved passes, 24 bits 25 ms more
            // than the one saved pass -- consecutive groups have to touch consecutive ranks, not just nearby ones)
            int ores = 0;
            BWTS_TRY(radix_sort_pairs(ctx, op, a, kb > 32 ? 32 : kb, &ores));
            {
                SpanGuard g(ctx, BWTS_K_RERANK, a, 20 * a);
                dg_unpack_kernel<<<dim3((unsigned)((a + 255) / 256)), dim3(256), 0, ctx->stream>>>(op.keys[ores], op.vals[ores], a, kb, sets[0].idx, sets[0].head);
                HIPC(hipGetLastError());
            }
            cur = sets[0];
            nxt = 1;
            // Opt-in (BWTS_DENSE_RUNS=1).  Exact, but measured slower on every text at hand (synthetic 282 -> 319 ms, real text
            // +7 %): groups' compositions change every few positions when three or more copies overlap, so runs are short, most
            // groups are activated within four rounds and then wait in the always-active rest like before -- while the flags,
            // ordinals and the partition cost 35 ms.  It pays on inputs that are two copies of one text.
            const bool runs_ok = [ctx] { const char *e = bwts_knob(ctx, "BWTS_DENSE_RUNS"); return e && atoi(e) == 1; }();
            if (runs_ok) {
                // activation rounds (see dg_runflags_kernel): flags -> group / run ordinals -> distance to the run's end -> stable
                // partition of the list by activation round.  Scratch: the rounds' working buffers, not in use yet.
                SpanGuard g(ctx, BWTS_K_RERANK, a, 60 * a);
                u8 *gflag = state;
                u32 *gord = t_idx, *rid = t_head, *gfirst = sets[1].idx;
                u64 *totals = cnt + 4;
                dg_runflags_kernel<CYCLIC><<<dim3((unsigned)((a + DG_OWN - 1) / DG_OWN)), dim3(DG_THREADS), 0, ctx->stream>>>(cur.idx, cur.head, a, n, d_fstart, k, gflag);
                DgRunIn rin{gflag};
                DgRunOut rout{gflag, gord, rid, gfirst, a, totals};
                BWTS_TRY((device_scan<true, u64>(ctx, a, rin, rout, OpAdd(), (u64)0, sp.scan_temp)));
                dg_actkeys_kernel<<<dim3((unsigned)((a + 255) / 256)), dim3(256), 0, ctx->stream>>>(cur.idx, cur.head, gord, rid, gfirst, a, totals, (u64)al.hstep,
                                                                                                   op.keys[0], op.vals[0]);
                HIPC(hipGetLastError());
                int pres = 0;
                BWTS_TRY(radix_sort_pairs(ctx, op, a, 8, &pres));
                dg_unpack8_kernel<<<dim3((unsigned)((a + 255) / 256)), dim3(256), 0, ctx->stream>>>(op.keys[pres], op.vals[pres], a, sets[0].idx, sets[0].head);
                HIPC(hipGetLastError());
                // where each activation round's share starts: row 0 of the pass's scanned tile table (digit-major offsets)
                u32 hb[256];
                HIPC(hipMemcpyAsync(hb, sp.tile_hist, sizeof(hb), hipMemcpyDeviceToHost, ctx->stream));
                HIPC(hipStreamSynchronize(ctx->stream));
                for (int r = 0; r < DG_MAX_ACT; r++) act_start[r] = hb[r];
                act_start[DG_MAX_ACT] = a;
                for (int r = 0; r < DG_MAX_ACT; r++) if (act_start[r] > act_start[r + 1]) return BWTS_E_INTERNAL;
                by_rounds = true;
                master = sets[0];
                cur = sets[1];
                nxt = 2;
                a = 0;                       // nothing is active yet: round 0 takes its share below
            }
        }
    }
    u64 act_round = 0;
    u64 rest_from = a0;                       // master elements from here on were never activated (only when the loop ends on "no split")
    // the step is quadrupled per round (three successor ranks per element) unless BWTS_DENSE_STEP=2 asks for plain doubling;
    // the activation rounds above are laid out for doubling
    const bool step4_ok = [ctx] { const char *e = bwts_knob(ctx, "BWTS_DENSE_STEP"); return !(e && atoi(e) == 2); }();
    const int nk = step4_ok && !by_rounds ? 3 : 1;
    for (u64 h = (u64)al.hstep;; h <<= (nk == 3 ? 2 : 1), act_round++) {
        rounds++;
        if (by_rounds && act_round < DG_MAX_ACT) {
            // this round's newly active groups join what earlier rounds left tied
            const u64 lo = act_start[act_round], cntb = act_start[act_round + 1] - lo;
            if (cntb) {
                HIPC(hipMemcpyAsync(cur.idx + a, master.idx + lo, cntb * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
                HIPC(hipMemcpyAsync(cur.head + a, master.head + lo, cntb * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
                a += cntb;
            }
        }
        const u64 waiting = by_rounds && act_round + 1 <= DG_MAX_ACT ? a0 - act_start[act_round + 1 < DG_MAX_ACT ? act_round + 1 : DG_MAX_ACT] : 0;
        if (a == 0) {
            if (waiting == 0) break;
            if (CYCLIC && rounds - 1 < BWTS_MAX_ROUND_STATS) ctx->tm.round_active[rounds - 1] = waiting;
            if (rounds > 80) return BWTS_E_INTERNAL;
            continue;                        // no group can split at this step; the next activation round comes with a larger one
        }
        HIPC(hipMemsetAsync(cnt, 0, (DG_CNT_BIG + DG_CNT_SPREAD) * sizeof(u64), ctx->stream));
        const u64 rtiles = (a + DG_OWN - 1) / DG_OWN;
        HIPC(hipMemsetAsync(tile_big + rtiles, 0, sizeof(u32), ctx->stream));      // the scan below turns counts into offsets; entry [rtiles] = total
        {
            SpanGuard g(ctx, BWTS_K_KEYBUILD, a, 29 * a);       // idx, head in; idx, head, state out; one rank gather
            if (nk == 3)
                dense_round_kernel<CYCLIC, 3><<<dim3((unsigned)rtiles), dim3(DG_THREADS), 0, ctx->stream>>>(
                    cur.idx, cur.head, a, sp.rank, n, h, d_fstart, k, t_idx, t_head, state, prev, out, (unsigned long long *)cnt, tile_big);
            else
                dense_round_kernel<CYCLIC, 1><<<dim3((unsigned)rtiles), dim3(DG_THREADS), 0, ctx->stream>>>(
                    cur.idx, cur.head, a, sp.rank, n, h, d_fstart, k, t_idx, t_head, state, prev, out, (unsigned long long *)cnt, tile_big);
            HIPC(hipGetLastError());
        }
        BWTS_TRY(read_small(ctx, SM_DGCNT, DG_CNT_BIG + DG_CNT_SPREAD));
        u64 m_big = 0;
        for (int c = 0; c < DG_CNT_SPREAD; c++) m_big += ctx->h_small[SM_DGCNT + DG_CNT_BIG + c];
        if (m_big > a) return BWTS_E_INTERNAL;
        const bool round_trace = [ctx] { const char *e = bwts_knob(ctx, "BWTS_ROUND_TRACE"); return e && atoi(e) == 1; }();
        if (round_trace) fprintf(stderr, "[rounds] round %u h %llu: list %llu, in larger groups %llu\n", rounds, (unsigned long long)h, (unsigned long long)a, (unsigned long long)m_big);
        if (m_big) {
            // larger groups: compact (with the successor ranks), radix sort by (group ordinal, successor rank), regroup, put back
            char *bb = nullptr;
            const size_t m8 = align_up((size_t)m_big * 8, 256), m4 = align_up((size_t)m_big * 4, 256);
            BWTS_TRY(aux_reserve_slot(ctx, 1, (nk == 3 ? 4 : 2) * m8 + (nk == 3 ? 4 : 3) * m4, &bb));
            // [first keys | second buffer] [positions | second buffer] [list slots]; quadrupled step: + [second keys] [sort buffer] [index buffer]
            u64 *bk[2] = {(u64 *)bb, (u64 *)(bb + m8)};
            u32 *bv[2] = {(u32 *)(bb + 2 * m8), (u32 *)(bb + 2 * m8 + m4)};
            u32 *bpos = (u32 *)(bb + 2 * m8 + 2 * m4);
            u64 *k23 = nk == 3 ? (u64 *)(bb + 2 * m8 + 3 * m4) : nullptr, *sk1 = nk == 3 ? (u64 *)(bb + 3 * m8 + 3 * m4) : nullptr;
            u32 *sv1 = nk == 3 ? (u32 *)(bb + 4 * m8 + 3 * m4) : nullptr;
            {
                SpanGuard g(ctx, BWTS_K_RERANK, m_big, 4 * (a / DG_OWN) + 30 * m_big);
                BWTS_TRY(exclusive_sum_u32(ctx, tile_big, rtiles + 1, sp.scan_temp));
                dg_big_collect_kernel<<<dim3((unsigned)rtiles), dim3(256), 0, ctx->stream>>>(state, a, tile_big, bpos);
                DgBigIn fin{bpos, t_head};
                // quadrupled step: the first sort works on (second key copy in bk[1], element index in bv[1])
                DgBigOut<CYCLIC> fout{bpos, t_head, t_idx, rb, sp.rank, n, h, d_fstart, k, bk[0], bv[0], k23, bk[1], bv[1]};
                BWTS_TRY((device_scan<false, u32>(ctx, m_big, fin, fout, OpAdd(), 0u, sp.scan_temp)));
            }
            int big_bits = bitlen_u64(m_big / (DG_CAP + 1)) + rb;          // ordinals < m_big / (DG_CAP + 1)
            if (big_bits > 64) return BWTS_E_RANGE;
            SortPlan bp;
            bp.tile_hist = sp.tile_hist; bp.scan_temp = sp.scan_temp;
            int rbig = 0;
            const u64 *sorted_k1 = nullptr;
            const u32 *src = nullptr, *positions = nullptr;
            if (nk == 3) {
                // LSD over two key words: stable sort by (rank at 2h, rank at 3h), then by (group ordinal, rank at h)
                bp.keys[0] = bk[1]; bp.keys[1] = sk1;
                bp.vals[0] = bv[1]; bp.vals[1] = sv1;
                int r1 = 0;
                BWTS_TRY(radix_sort_pairs(ctx, bp, m_big, 2 * rb, &r1));
                u64 *kin = bp.keys[r1], *kout = bp.keys[r1 ^ 1];
                u32 *vin = bp.vals[r1], *vout = bp.vals[r1 ^ 1];
                {
                    SpanGuard g(ctx, BWTS_K_RERANK, m_big, 20 * m_big);
                    dg_stage2_keys_kernel<<<dim3((unsigned)((m_big + 255) / 256)), dim3(256), 0, ctx->stream>>>(vin, bk[0], m_big, kin);
                    HIPC(hipGetLastError());
                }
                bp.keys[0] = kin; bp.keys[1] = kout;
                bp.vals[0] = vin; bp.vals[1] = vout;
                BWTS_TRY(radix_sort_pairs(ctx, bp, m_big, big_bits, &rbig));
                sorted_k1 = bp.keys[rbig]; src = bp.vals[rbig]; positions = bv[0];
            } else {
                bp.keys[0] = bk[0]; bp.keys[1] = bk[1];
                bp.vals[0] = bv[0]; bp.vals[1] = bv[1];
                BWTS_TRY(radix_sort_pairs(ctx, bp, m_big, big_bits, &rbig));
                sorted_k1 = bk[rbig]; positions = bv[rbig];
            }
            {
                SpanGuard g(ctx, BWTS_K_RERANK, m_big, 36 * m_big);
                DgRegroupIn rin{sorted_k1, m_big, rb, src, k23};
                DgRegroupOut rout{sorted_k1, positions, bpos, m_big, rb, t_idx, t_head, state, prev, out, (unsigned long long *)cnt, src, k23};
                BWTS_TRY((device_scan<true, u64>(ctx, m_big, rin, rout, OpMax2(), (u64)0, sp.scan_temp)));
            }
        }
        {
            SpanGuard g(ctx, BWTS_K_RERANK, a, 18 * a);
            const u64 tiles = (a + DC_TILE - 1) / DC_TILE;
            u32 *partial = (u32 *)sp.scan_temp;
            dg_count_kernel<<<dim3((unsigned)tiles), dim3(DC_THREADS), 0, ctx->stream>>>(state, a, partial);
            BWTS_TRY((device_scan_partials<u32, OpAdd>(ctx, tiles, OpAdd(), 0u, sp.scan_temp)));
            dg_compact_kernel<<<dim3((unsigned)tiles), dim3(DC_THREADS), 0, ctx->stream>>>(state, t_idx, t_head, a, partial, sp.rank, sets[nxt].idx,
                                                                                          sets[nxt].head, cnt + 0);
            HIPC(hipGetLastError());
        }
        BWTS_TRY(read_small(ctx, SM_DGCNT, 4));
        const u64 a_new = ctx->h_small[SM_DGCNT + 0];
        const u64 splits = ctx->h_small[SM_DGCNT + 2];
        if (a_new > a) return BWTS_E_INTERNAL;
        cur = sets[nxt];
        nxt = by_rounds ? (nxt == 1 ? 2 : 1) : nxt ^ 1;
        a = a_new;
        if (CYCLIC && rounds - 1 < BWTS_MAX_ROUND_STATS) ctx->tm.round_active[rounds - 1] = a + waiting;
        if (a == 0 && waiting == 0) break;
        // no group split -- the groups still waiting for their activation round provably do not split either -- so the partition is
        // stable under doubling: what is left are groups of equal infinite words
        if (CYCLIC && splits == 0) { rest_from = by_rounds ? act_start[act_round + 1 < DG_MAX_ACT ? act_round + 1 : DG_MAX_ACT] : a0; break; }
        if (!CYCLIC && h >= n) return BWTS_E_INTERNAL;  // suffixes are distinct; cannot happen
        if (rounds > 80) return BWTS_E_INTERNAL;
    }
    if (need_sa) {
        SpanGuard g(ctx, BWTS_K_RERANK, n, 8 * n);
        u64 blocks = (n + 255) / 256; if (blocks > 16384) blocks = 16384;
        sa_from_rank_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(sp.rank, n, SA);
        HIPC(hipGetLastError());
    }
    if (a) {
        // groups of equal infinite words: their members take the group's slots in list order
        SpanGuard g(ctx, BWTS_K_EMIT, a, 10 * a);
        DgRestIn rin{cur.head};
        DgRestOut rout{cur.idx, cur.head, prev, out, need_sa ? SA : nullptr};
        BWTS_TRY((device_scan<true, u32>(ctx, a, rin, rout, OpMax(), 0u, sp.scan_temp)));
    }
    if (by_rounds && rest_from < a0) {
        // ... and so do the groups that were still waiting for their activation round
        const u64 m = a0 - rest_from;
        SpanGuard g(ctx, BWTS_K_EMIT, m, 10 * m);
        DgRestIn rin{master.head + rest_from};
        DgRestOut rout{master.idx + rest_from, master.head + rest_from, prev, out, need_sa ? SA : nullptr};
        BWTS_TRY((device_scan<true, u32>(ctx, m, rin, rout, OpMax(), 0u, sp.scan_temp)));
    }
    *rounds_io = rounds;
    return BWTS_OK;
}
