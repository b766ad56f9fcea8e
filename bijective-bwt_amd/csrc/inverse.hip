// inverse.hip -- BWTS inverse transform on the GPU.
//
// Replaces the inline core of /root/reference/unbwts.c:31-86:
//   :34-43  histogram + exclusive scan -> row 0 of the scanned [tile][symbol] table (ctab_from_tiles_kernel)
//   :50-52  prev[i]=counts[B[i]]++ (stable LF map) -> lf_hist_kernel + column scan (radix.hip) + lf_rank_kernel
//   :66-86  cycle walk, smallest unvisited index first, text written backwards
//           -> splitter walk recording every segment's symbols, reduced-list ranking by pointer jumping,
//              placement of the recorded segments
// The reference follows ONE cycle at a time (n dependent loads).  Here every G-th index is a splitter; a lane walks
// from its splitter to the next one, so ~n/G walks run concurrently, and LF is chased exactly once.  Cycles that contain
// no splitter are found from the walk's index log (or, in the alternative modes, from visited marks) and resolved
// separately.
#include "internal.h"
#include "device_utils.h"
#include "scan_templ.h"

#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#define LF_THREADS 256
#define LF_WAVES   4
#define LF_ITEMS   16
#define LF_TILE    (LF_THREADS * LF_ITEMS)
#define LF_VISITED 0xffffffffu     // written over an entry once the walk has read it (LF is chased exactly once)


#define SMI_COUNTERS 320

// ------------------------------------------------------------------------------------
// stable LF map
// ------------------------------------------------------------------------------------
// per-tile symbol counts.  The order inside the tile does not matter here: a thread takes 16 consecutive bytes (one
// 16-byte load when the tile is aligned) and counts into one of 16 lane-interleaved copies of the bins (skewed text would
// otherwise serialise a wave on its most frequent symbols).
#define LFH_COPIES 16
__global__ __launch_bounds__(LF_THREADS) void lf_hist_kernel(const u8 *__restrict__ B, u64 n, u32 *__restrict__ tile_hist)
{
    __shared__ u32 bins[LFH_COPIES][256];
    const int tid = threadIdx.x;
    u32 *mine = bins[tid & (LFH_COPIES - 1)];
    for (int i = tid; i < LFH_COPIES * 256; i += LF_THREADS) ((u32 *)bins)[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * LF_TILE;
    const u64 i0 = base + (u64)tid * LF_ITEMS;
    static_assert(LF_ITEMS == 16, "one 16-byte load per thread");
    if (i0 + LF_ITEMS <= n && (((uintptr_t)B + i0) & 15) == 0) {
        const uint4 q = *(const uint4 *)(B + i0);
        const u32 ws[4] = {q.x, q.y, q.z, q.w};
        // (a transform of text is made of runs: a thread adds each run of its 16 bytes once: the inverse of 1 GiB of real text 33.3 -> 32.0 ms)
        u32 cur = ws[0] & 255u, cnt = 0;
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
                const u32 b = (ws[a] >> (8 * bb)) & 255u;
                if (b == cur) cnt++;
                else { atomicAdd(&mine[cur], cnt); cur = b; cnt = 1; }
            }
        }
        atomicAdd(&mine[cur], cnt);
    } else {
        for (int j = 0; j < LF_ITEMS; j++) if (i0 + j < n) atomicAdd(&mine[B[i0 + j]], 1u);
    }
    __syncthreads();
    u32 s = 0;
#pragma unroll
    for (int c = 0; c < LFH_COPIES; c++) s += bins[c][tid];
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

// C[c] = first slot of symbol c = tile 0's offset in the scanned [tile][symbol] table; C[256] = n.  The table is 32-bit:
// a boundary equal to 2^32 (n = 2^32, symbols above the largest one present) reads as a value below its predecessor --
// unless the predecessor is 0 too, i.e. one symbol holds all 2^32 positions: then every entry reads 0 and B[0] says which
// symbol that is.
__global__ void ctab_from_tiles_kernel(const u32 *__restrict__ tile_off, u64 n, const u8 *__restrict__ B, u64 *__restrict__ C)
{
    if (threadIdx.x != 0) return;
    u64 prev = 0;
    bool all_zero = true;
    for (int c = 0; c < 256; c++) {
        u64 v = tile_off[c];
        all_zero = all_zero && v == 0;
        if (v < prev) v += 0x100000000ull;
        C[c] = v;
        prev = v;
    }
    if (all_zero && n == 0x100000000ull) {
        const int b = B[0];
        for (int c = b + 1; c < 256; c++) C[c] = n;
    }
    C[256] = n;
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

// A lane walks LF from its splitter to the next one.  Along the way it leaves evidence of the entries it visits
// (MARK below), keeps the smallest index seen, and records the symbols it passes -- B[x] read off LF[x] -- into
// its node's slot of `seg`, 4 at a time.  A walk that reaches `slot` steps without meeting a splitter closes its
// node there and continues as a fresh virtual node (ids >= s, handed out by an atomic counter), so no segment
// outgrows its slot.  A wave pulls batches of splitter ids from a shared counter and hands them to its lanes as
// they finish (one atomic per WALK_BATCH walks); every walk ends at the next splitter (LF is a permutation) and
// a lane that finds the counter exhausted stops asking, so every wave drains.
#ifndef WALK_BATCH
#define WALK_BATCH 128
#endif
// MARK: how the walk leaves evidence of the entries it visited, so that cycles without a splitter can be found:
//   0  index log   -- every step the wave appends the indices its active lanes stand on to its own log chunk, back to
//                     back (one contiguous store per wave and step; the log is a set, it does not say which node an
//                     index belongs to).  The unvisited complement comes from bucket_indices_kernel +
//                     unvisited_from_buckets_kernel.  The walk then costs one random line fill per step and nothing
//                     else that is random: measured 58 -> 35 ms at n = 2^30 against sentinel marks.
//   1  sentinel    -- the entry is overwritten with LF_VISITED (one extra random 32-byte write per step)
//   2  byte map    -- marks[x] = 1 (when LF_VISITED could be a real entry: n = 2^32)
// WALK_PROFILE (compile-time) stamps wall-clock times of first start / pool exhausted / wave ends into ticket[8..12]:
// at n = 2^30 the pool runs dry at ~29 ms whatever the walker count (64 K .. 512 K lanes) and batch size, and the
// longest residual chain (~G ln(walkers) dependent steps) adds ~5 ms; only a smaller G shortens that tail.
#define IDX_RANGE_LOG2 20
#define IDX_MAX_BUCKETS 4096
#define IDX_THREADS 512
#define IDX_PER_THREAD 32
#define IDX_CHUNK (IDX_THREADS * IDX_PER_THREAD)      // entries of one log chunk (= one tile of bucket_indices_kernel)
#define IDX_FILL_STRIDE 32            // u32 words between two bucket counters (128 bytes)
#define MARK_LOG 0
#define MARK_SENTINEL 1
#define MARK_BYTEMAP 2
#define MARK_MOMENTS 3
//   3  moments     -- nothing is written per step: the workgroup keeps, per residue class of the index mod 2^10 (2^12 above n = 2^30), how many
//                     indices it visited, the sum of their quotients and the sum of the squares (three LDS atomics).  A class that
//                     misses one or two indices names them by arithmetic; a class that misses more is searched element by element
//                     (moments_chase_kernel: an element is unreached iff its own chase returns to it before it meets a splitter).
//                     (Classes by the high bits -- ranges -- were the first form: the few dozen unreached elements of a text are the
//                     rotations of its last, small Lyndon factors and sit in a handful of ranges, a dozen to each.)
//                     The micro-benchmark puts the log at 7 % of the walk, and its scan at 1.4 ms (tools/micro/walk_steps.hip).
#define MOM_LOG2_SMALL 10                    // classes: 2^10 up to n = 2^30 (20 KB of LDS), 2^12 above (80 KB: a class stays at 2^20 elements)
#define MOM_LOG2_LARGE 12
#define MOM_MAX_BUCKETS (1u << MOM_LOG2_LARGE)
template <int MARK, int SBW = 16 /* registers of recorded symbols per store: 16 = 64-byte blocks, 4 = 16-byte ones (BWTS_WALK_SYMS=16) */>
__global__ __launch_bounds__(256) void walk_record_kernel(u32 *__restrict__ LF, u8 *__restrict__ marks, u32 *__restrict__ idxlog, u64 s, u64 node_cap, int g, u32 slot,
                                                          const u64 *__restrict__ Cg, u8 *__restrict__ seg,
                                                          uint4 *__restrict__ noderec /* x next node, y segment length, z smallest element, w its offset */,
                                                          unsigned long long *__restrict__ ticket,
                                                          unsigned long long *__restrict__ vcount,
                                                          unsigned long long *__restrict__ overflow,
                                                          unsigned long long *__restrict__ chunk_ctr, u32 *__restrict__ chunk_fill, u64 log_chunks,
                                                          u32 nbuckets, u32 *__restrict__ bucket_seen,
                                                          int mom_shift = 0, unsigned long long *__restrict__ mom = nullptr /* [3][2^mom_shift]: counts, sums, sums of squares */)
{
    __shared__ u64 Ctab[257];
    extern __shared__ __attribute__((aligned(16))) unsigned long long walk_mom_sm[];     // MARK_MOMENTS: 2^mom_shift sums, sums of squares, counts
    const u32 mom_classes = MARK == MARK_MOMENTS ? 1u << mom_shift : 0u;
    unsigned long long *msum = walk_mom_sm, *msq = walk_mom_sm + mom_classes;
    u32 *mcnt = (u32 *)(walk_mom_sm + 2 * mom_classes);
    if (MARK == MARK_MOMENTS) for (u32 b = threadIdx.x; b < mom_classes; b += 256) { mcnt[b] = 0; msum[b] = 0; msq[b] = 0; }
    // (Round 3 tried two things here and measured both worse: fetching the next LF entry before the current step's symbol search and
    // moments -- the walk is bound by the memory system's line-fill rate, not by a lane's dependency chain, and the second read in
    // flight per lane only lengthens the queues: dna 2^32 119.7 -> 129.5 ms, zipf 2^30 27.4 -> 27.8 ms -- and count + sum in one
    // 64-bit LDS atomic, which changed nothing measurable.  tools/sessions/r03y.sh, r03af.sh.)
    // MARK_LOG: how many indices of each 2^IDX_RANGE_LOG2-range this workgroup visited.  A range that ends up with all of
    // its indices counted holds nothing unvisited, and its log entries need not be looked at again.
    __shared__ u32 bseen[MARK == MARK_LOG ? IDX_MAX_BUCKETS : 1];
    if (MARK == MARK_LOG) for (u32 b = threadIdx.x; b < nbuckets; b += 256) bseen[b] = 0;
    for (int i = threadIdx.x; i < 257; i += 256) Ctab[i] = Cg[i];
    __syncthreads();
    const u32 gmask = (1u << g) - 1u;
    bool have = false, done = false;
    u64 my = 0;
    u32 x = 0, len = 0, mn = 0, mnoff = 0;
    // 64 recorded symbols wait in 16 registers for one 64-byte store into the node's slot (four 16-byte words at once): with 16 symbols
    // per store the walk ran at 36 G steps/s, with 64 at 40 (tools/micro/walk_steps.hip: the scattered 16-byte stores cost 19 % of a bare chase,
    // the 64-byte ones 3 %)
    u32 sb[SBW];
    constexpr u32 SBM = 4 * SBW - 1;           // symbols per block - 1
#pragma unroll
    for (int q = 0; q < SBW; q++) sb[q] = 0;
    u64 bnext = 0, bend = 0;            // the wave's current batch of splitter ids (wave-uniform)
    bool exhausted = false;
#ifdef WALK_PROFILE
    unsigned long long *prof = ticket + 8;
    if (lane_id() == 0) atomicMin(&prof[0], (unsigned long long)wall_clock64());
    bool stamped = false;
#endif
    u64 lcur = 0, lbase = 0, lend = 0;  // MARK_LOG: the wave's open log chunk [lbase, lend) and its fill cursor (wave-uniform)
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
#ifdef WALK_PROFILE
                if (exhausted && !stamped) { stamped = true; if (lane_id() == leader) { atomicMin(&prof[1], (unsigned long long)wall_clock64()); atomicMax(&prof[3], (unsigned long long)wall_clock64()); } }
#endif
            }
            if (!have && !done) {
                const u64 id = bnext + (u64)__popcll(need & lanemask_lt());
                if (id < bend) {
                    have = true; my = id; x = (u32)(my << g); len = 0; mn = x; mnoff = 0;
#pragma unroll
                    for (int q = 0; q < SBW; q++) sb[q] = 0;
                }
                else if (exhausted) done = true;      // no work left anywhere: this lane never asks again
            }
            const u64 taken = bnext + (u64)__popcll(need);
            bnext = taken < bend ? taken : bend;
        }
        if (__ballot(have || !done) == 0) break;     // every lane has seen the counter run dry
        if (MARK == MARK_LOG) {
            // the indices the wave visits in this step go to its log back to back: one contiguous store per wave
            const u64 act = __ballot(have);
            const u32 na = (u32)__popcll(act);
            if (na) {
                if (lcur + na > lend) {
                    const int leader = __ffsll((unsigned long long)act) - 1;
                    unsigned long long cid = 0;
                    if (lane_id() == leader) {
                        if (lend) chunk_fill[lbase / IDX_CHUNK] = (u32)(lcur - lbase);
                        cid = atomicAdd(chunk_ctr, 1ull);
                        if (cid >= log_chunks) { atomicAdd(overflow, 1ull); cid = log_chunks - 1; }   // cannot happen (host sizes the log); stay in bounds
                    }
                    cid = shfl_t((u64)cid, leader);
                    lbase = lcur = cid * IDX_CHUNK;
                    lend = lbase + IDX_CHUNK;
                }
                if (have) { idxlog[lcur + (u64)__popcll(act & lanemask_lt())] = x; atomicAdd(&bseen[x >> IDX_RANGE_LOG2], 1u); }
                lcur += na;
            }
        }
        if (MARK == MARK_MOMENTS && have) {
            const u32 b = x & (mom_classes - 1u);                           // classes by the low bits: the unreached elements of real inputs cluster in rank
            const unsigned long long o = x >> mom_shift;
            atomicAdd(&mcnt[b], 1u); atomicAdd(&msum[b], o); atomicAdd(&msq[b], o * o);
        }
        if (have) {
            const u32 y = LF[x];
            if (MARK == MARK_BYTEMAP) marks[x] = 1;
            else if (MARK == MARK_SENTINEL) LF[x] = LF_VISITED;       // the entry is not needed again
            {
                const u32 sh = symbol_of(Ctab, y) << (8 * (len & 3u));
                const u32 w = (len >> 2) & (u32)(SBW - 1);
#pragma unroll
                for (int q = 0; q < SBW; q++) sb[q] |= w == (u32)q ? sh : 0u;
            }
            if ((len & SBM) == SBM) {                        // (only reached when the slot holds at least a block)
                uint4 *d = (uint4 *)(seg + my * slot + (len & ~SBM));
#pragma unroll
                for (int q = 0; q < SBW / 4; q++) d[q] = make_uint4(sb[4 * q], sb[4 * q + 1], sb[4 * q + 2], sb[4 * q + 3]);
#pragma unroll
                for (int q = 0; q < SBW; q++) sb[q] = 0;
            }
            len++;
            x = y;
            const bool at_splitter = (x & gmask) == 0;
            if (at_splitter || len == slot) {
                if (len & SBM) {                                                                             // slot is a multiple of 16
                    uint4 *d = (uint4 *)(seg + my * slot + (len & ~SBM));
                    const u32 rem = len & SBM;
#pragma unroll
                    for (int q = 0; q < SBW / 4; q++) if ((u32)q * 16u < rem) d[q] = make_uint4(sb[4 * q], sb[4 * q + 1], sb[4 * q + 2], sb[4 * q + 3]);
                }
                u64 next_node;
                if (at_splitter) {
                    next_node = x >> g;
                    have = false;
                } else {
                    next_node = s + atomicAdd(vcount, 1ull);
                    if (next_node >= node_cap) { atomicAdd(overflow, 1ull); next_node = node_cap - 1; }
                }
                noderec[my] = make_uint4((u32)next_node, len, mn, mnoff);
                if (!at_splitter) {
                    my = next_node; len = 0; mn = x; mnoff = 0;
#pragma unroll
                    for (int q = 0; q < SBW; q++) sb[q] = 0;
                }
            } else if (x < mn) { mn = x; mnoff = len; }
        }
    }
    if (MARK == MARK_LOG && lend && lane_id() == 0) chunk_fill[lbase / IDX_CHUNK] = (u32)(lcur - lbase);
    if (MARK == MARK_LOG) {
        __syncthreads();                  // every wave leaves the loop (the pool runs dry for all of them)
        for (u32 b = threadIdx.x; b < nbuckets; b += 256) { const u32 c = bseen[b]; if (c) atomicAdd(&bucket_seen[b], c); }
    }
    if (MARK == MARK_MOMENTS) {
        __syncthreads();
        for (u32 b = threadIdx.x; b < mom_classes; b += 256) {
            const u32 c = mcnt[b];
            if (c) { atomicAdd(&mom[b], (unsigned long long)c); atomicAdd(&mom[mom_classes + b], msum[b]); atomicAdd(&mom[2 * mom_classes + b], msq[b]); }
        }
    }
#ifdef WALK_PROFILE
    if (lane_id() == 0) { atomicMax(&prof[2], (unsigned long long)wall_clock64()); atomicMin(&prof[4], (unsigned long long)wall_clock64()); }
#endif
}

// out[end_c - t] = B[LF^t(min_c)] (unbwts.c:73-82): node v's recorded symbols go to out[opos - i], wrapping to the
// cycle's end once the walk passes the cycle's smallest element.  A thread moves 16 symbols at a time: one aligned
// 16-byte load from the node's slot, bytes reversed in registers, one 16-byte store (the destination is not aligned in
// general: the store goes through a packed type, so the compiler picks what the target allows).  The one chunk of a
// node that straddles the wrap point, and a ragged tail, go byte by byte.
struct __attribute__((packed, aligned(1))) Unaligned16 { u32 w[4]; };
__global__ __launch_bounds__(256) void place_segments_kernel(const u8 *__restrict__ seg, u64 nodes, u32 slot, int tpn_log2,
                                                             const uint4 *__restrict__ noderec, const u32 *__restrict__ opos,
                                                             const u32 *__restrict__ wrap_at, const u32 *__restrict__ cyc_len,
                                                             u8 *__restrict__ out)
{
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 v = gid >> tpn_log2;
    if (v >= nodes) return;
    const u32 sub = (u32)(gid & ((1ull << tpn_log2) - 1ull)), tpn = 1u << tpn_log2;
    const u32 len = noderec[v].y, o = opos[v], wr = wrap_at[v], L = cyc_len[v];
    const u8 *src = seg + v * slot;
    for (u32 c = sub * 16; c < len; c += tpn * 16) {
        if (c + 16 <= len && (c + 16 <= wr || c >= wr)) {
            const uint4 q = *(const uint4 *)(src + c);
            // symbols c .. c+15 land on out[base - 15 .. base], last symbol first
            const u64 base = c >= wr ? (u64)o - c + L : (u64)o - c;
            Unaligned16 r;
            r.w[0] = __builtin_bswap32(q.w); r.w[1] = __builtin_bswap32(q.z);
            r.w[2] = __builtin_bswap32(q.y); r.w[3] = __builtin_bswap32(q.x);
            *(Unaligned16 *)(out + base - 15) = r;
        } else {
            const u32 e = c + 16 < len ? c + 16 : len;
            for (u32 i = c; i < e; i++) out[i >= wr ? (u64)o - i + L : (u64)o - i] = src[i];
        }
    }
}

// ------------------------------------------------------------------------------------
// elements no walk reached (cycles without a splitter): one sweep, wave-aggregated append
// ------------------------------------------------------------------------------------
template <int MARK>
__global__ __launch_bounds__(256) void collect_unvisited_kernel(const u32 *__restrict__ LF, const u8 *__restrict__ marks, u64 n,
                                                                u32 *__restrict__ uidx, u32 *__restrict__ ulf, u64 cap,
                                                                unsigned long long *__restrict__ count)
{
    for (u64 base = (u64)blockIdx.x * 256; base < n; base += (u64)gridDim.x * 256) {
        const u64 i = base + threadIdx.x;
        bool un = false;
        u32 v = 0;
        if (i < n) {
            if (MARK == MARK_BYTEMAP) { un = marks[i] == 0; if (un) v = LF[i]; }
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

// ---- index-log mode: which indices did no walk log? ------------------------------------------------------------------
// Step 1: the logged indices are appended to buckets of 2^IDX_RANGE_LOG2 consecutive index values.  A workgroup takes
// one log chunk (its first chunk_fill[] entries): counts per bucket in LDS, reserves room in every bucket the chunk
// touches (one global atomic each; the counters sit on separate cache lines), orders the chunk by bucket in LDS and
// copies it out, so a bucket's share leaves as one contiguous run instead of one request per entry.
// Step 2: one workgroup per bucket sets a bit per logged index in an LDS bitmap and reports the zero bits.
// Step 0: the walk counted the visited indices of every bucket; only DEFICIENT buckets (count < size) can hold an unvisited
// index, and only their log entries take part in steps 1 and 2 (natural inputs: a few dozen unvisited elements, so a
// few percent of the buckets).
__global__ __launch_bounds__(256) void bucket_deficit_kernel(const u32 *__restrict__ bucket_seen, u32 nbuckets, u64 n, u32 *__restrict__ deficient)
{
    const u32 b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbuckets) return;
    const u64 lo = (u64)b << IDX_RANGE_LOG2;
    const u64 size = n - lo < (1ull << IDX_RANGE_LOG2) ? n - lo : (1ull << IDX_RANGE_LOG2);
    deficient[b] = (u64)bucket_seen[b] < size ? 1u : 0u;
}

static inline size_t bucket_indices_lds_bytes(u32 nbuckets) { return (size_t)IDX_CHUNK * 4 + 3 * (size_t)nbuckets * 4; }
__global__ __launch_bounds__(IDX_THREADS) void bucket_indices_kernel(const u32 *__restrict__ idxlog, const u32 *__restrict__ chunk_fill,
                                                                     u32 nbuckets, const u32 *__restrict__ deficient,
                                                                     u32 *__restrict__ bucket_fill, u32 *__restrict__ bucket_data)
{
    extern __shared__ __attribute__((aligned(16))) u32 idx_lds[];
    u32 *sorted = idx_lds;                    // IDX_CHUNK entries ordered by bucket
    u32 *cnt = idx_lds + IDX_CHUNK;           // per bucket: count, then the fill cursor within `sorted`
    u32 *delta = cnt + nbuckets;              // per bucket: (reserved offset in the bucket) - (start in `sorted`)
    u32 *defl = delta + nbuckets;             // per bucket: takes part?
    __shared__ u32 scan_sm[IDX_THREADS / 64];
    const int tid = threadIdx.x;
    const u32 clen = chunk_fill[blockIdx.x];
    if (clen == 0) return;
    const u32 *src = idxlog + (u64)blockIdx.x * IDX_CHUNK;
    u32 x[IDX_PER_THREAD];
#pragma unroll
    for (int q = 0; q < IDX_PER_THREAD; q++) {
        const u32 i = (u32)q * IDX_THREADS + tid;
        x[q] = i < clen ? src[i] : 0u;
    }
    for (u32 b = tid; b < nbuckets; b += IDX_THREADS) { cnt[b] = 0; defl[b] = deficient[b]; }
    __syncthreads();
    u32 keep = 0;                             // bit q: entry q exists and its bucket takes part
#pragma unroll
    for (int q = 0; q < IDX_PER_THREAD; q++)
        if ((u32)q * IDX_THREADS + tid < clen && defl[x[q] >> IDX_RANGE_LOG2]) { keep |= 1u << q; atomicAdd(&cnt[x[q] >> IDX_RANGE_LOG2], 1u); }
    if (__syncthreads_or(keep != 0) == 0) return;       // nothing of this chunk lies in a deficient bucket
    // exclusive scan of the counts -> start of each bucket's run in `sorted`; room in the bucket from a global atomic
    u32 kept = 0;                             // entries of this chunk that take part
    {
        const u32 per = (nbuckets + IDX_THREADS - 1) / IDX_THREADS;      // buckets a thread scans (<= 8)
        u32 c[8], run = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 b = (u32)tid * per + j;
            c[j] = (u32)j < per && b < nbuckets ? cnt[b] : 0u;
            run += c[j];
        }
        u32 exc = block_scan_exclusive<u32, OpAdd, IDX_THREADS / 64>(run, OpAdd(), 0u, scan_sm, &kept);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 b = (u32)tid * per + j;
            if ((u32)j < per && b < nbuckets) {
                const u32 g = c[j] ? atomicAdd(&bucket_fill[(size_t)b * IDX_FILL_STRIDE], c[j]) : 0u;
                cnt[b] = exc;
                delta[b] = g - exc;
                exc += c[j];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IDX_PER_THREAD; q++)
        if ((keep >> q) & 1u) sorted[atomicAdd(&cnt[x[q] >> IDX_RANGE_LOG2], 1u)] = x[q];
    __syncthreads();
    for (u32 i = tid; i < kept; i += IDX_THREADS) {
        const u32 v = sorted[i], b = v >> IDX_RANGE_LOG2;
        bucket_data[((u64)b << IDX_RANGE_LOG2) + (u32)(delta[b] + i)] = v;
    }
}

__global__ __launch_bounds__(1024) void unvisited_from_buckets_kernel(const u32 *__restrict__ deficient, const u32 *__restrict__ bucket_fill,
                                                                      const u32 *__restrict__ bucket_data,
                                                                      const u32 *__restrict__ LF, u64 n, u32 *__restrict__ uidx,
                                                                      u32 *__restrict__ ulf, u64 cap, unsigned long long *__restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) u32 bm[];      // 2^IDX_RANGE_LOG2 bits = 128 KB (dynamic: above the static limit)
    const u32 b = blockIdx.x;
    if (!deficient[b]) return;
    for (u32 w = threadIdx.x; w < (1u << IDX_RANGE_LOG2) / 32; w += 1024) bm[w] = 0;
    __syncthreads();
    const u32 fill = bucket_fill[(size_t)b * IDX_FILL_STRIDE];
    const u32 *src = bucket_data + ((u64)b << IDX_RANGE_LOG2);
    for (u32 i = threadIdx.x; i < fill; i += 1024) {
        const u32 o = src[i] & ((1u << IDX_RANGE_LOG2) - 1u);
        atomicOr(&bm[o >> 5], 1u << (o & 31u));
    }
    __syncthreads();
    // natural inputs leave a few dozen zero bits; constant or sorted inputs leave nearly all of them: the room in the
    // list is reserved once per wave and step, not once per element
    const u64 lo = (u64)b << IDX_RANGE_LOG2;
    for (u32 w0 = 0; w0 < (1u << IDX_RANGE_LOG2) / 32; w0 += 1024) {
        const u32 w = w0 + threadIdx.x;
        u32 zeros = ~bm[w];
        const u64 xbase = lo + (u64)w * 32;
        if (xbase >= n) zeros = 0;
        else if (n - xbase < 32) zeros &= (1u << (u32)(n - xbase)) - 1u;
        const u32 mine = (u32)__popc(zeros);
        const u32 inc = wave_scan_inclusive(mine, OpAdd());
        const u32 tot = shfl_t(inc, 63);
        if (tot == 0) continue;
        unsigned long long base = 0;
        if (lane_id() == 63) base = atomicAdd(count, (unsigned long long)tot);
        base = shfl_t((u64)base, 63);
        u64 at = base + (inc - mine);
        while (zeros) {
            const u32 bit = (u32)__ffs((int)zeros) - 1u;
            zeros &= zeros - 1u;
            const u64 x = xbase + bit;
            if (at < cap) { uidx[at] = (u32)x; ulf[at] = LF[x]; }
            at++;
        }
    }
}

// ------------------------------------------------------------------------------------
// reduced-list ranking (splitter nodes): two levels, linear work
// ------------------------------------------------------------------------------------
// node v (noderec[v]): x = next node on its cycle, y = elements in its segment, z = smallest element of the segment,
// w = that element's distance from the node's first element.  Wanted per node: where its segment goes in the text, i.e.
// its cycle's end position, the cycle's length, and the node's distance from the cycle's smallest element.
//
// Pointer jumping over all nodes is O(s log s) work in 2 log s launches; that is what kept the splitter spacing G large.
// Instead every L2_H-th node is a level-2 splitter: a lane walks the node list from its level-2 splitter to the next one
// (lr2_walk_kernel: sums, minima, visit marks), nodes no such walk reached -- node cycles without a level-2 splitter --
// join the level-2 list as they are (lr2_collect / lr2_fill), the level-2 list (~ s / L2_H entries) is ranked by pointer
// jumping, and a second walk hands the positions down to the nodes (lr2_distribute_kernel).
#define L2_H 32
#define LR_NIL 0xffffffffu

__global__ __launch_bounds__(256) void lr2_walk_kernel(const uint4 *__restrict__ noderec, u64 s2, u8 *__restrict__ visited, uint4 *__restrict__ rec2)
{
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    if (w >= s2) return;
    u32 v = (u32)(w * L2_H), acc = 0, mn = 0, mnoff = 0;
    bool first = true;
    do {
        const uint4 r = noderec[v];
        visited[v] = 1;
        if (first || r.z < mn) { mn = r.z; mnoff = acc + r.w; first = false; }
        acc += r.y;
        v = r.x;
    } while (v % L2_H != 0);
    rec2[w] = make_uint4(v / L2_H, acc, mn, mnoff);
}

// nodes on cycles without a level-2 splitter: each becomes a level-2 entry of its own (id = s2 + position in U)
__global__ __launch_bounds__(256) void lr2_collect_kernel(const u8 *__restrict__ visited, u64 s_all, u64 s2, u32 *__restrict__ U,
                                                          u32 *__restrict__ id2of, unsigned long long *__restrict__ count)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    const bool un = v < s_all && !visited[v];
    const u64 m = __ballot(un);
    if (m == 0) return;
    const int leader = __ffsll((unsigned long long)m) - 1;
    unsigned long long b = 0;
    if (lane_id() == leader) b = atomicAdd(count, (unsigned long long)__popcll(m));
    b = shfl_t((u64)b, leader);
    if (un) {
        const u64 at = b + (u64)__popcll(m & lanemask_lt());
        U[at] = (u32)v;
        id2of[v] = (u32)(s2 + at);
    }
}
__global__ __launch_bounds__(256) void lr2_fill_kernel(const u32 *__restrict__ U, u64 nu2, u64 s2, const uint4 *__restrict__ noderec,
                                                       const u32 *__restrict__ id2of, uint4 *__restrict__ rec2)
{
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j >= nu2) return;
    const uint4 r = noderec[U[j]];
    rec2[s2 + j] = make_uint4(id2of[r.x], r.y, r.z, r.w);       // the successor of an unreached node is unreached too
}

// ---- pointer jumping over the level-2 list -------------------------------------------------------------------------
// A round reads the record of the entry it hops to: the fields travel together (16 and 8 bytes), so a round costs one
// random line fill per entry instead of three (two).
//   LrMin  x = smallest entry id seen (-> leader), y = smallest element seen, z = hop target
//   LrSum  x = segment lengths summed up to the cut, y = hop target (LR_NIL at the cut)
typedef uint4 LrMin;
typedef uint2 LrSum;

__global__ __launch_bounds__(256) void lr_init_kernel(u64 s, const uint4 *__restrict__ rec2, LrMin *__restrict__ rec)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v < s) { const uint4 r = rec2[v]; rec[v] = make_uint4((u32)v, r.z, r.x, 0u); }
}

// after r rounds an entry has folded in the 2^r entries that follow it
__global__ __launch_bounds__(256) void lr_jump_min_kernel(u64 s, const LrMin *__restrict__ in, LrMin *__restrict__ out)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const LrMin a = in[v];
    const LrMin b = in[a.z];
    out[v] = make_uint4(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, b.z, 0u);
}

// cut every cycle in front of its leader, then suffix sums of the lengths
__global__ __launch_bounds__(256) void lr_cut_kernel(u64 s, const uint4 *__restrict__ rec2, const LrMin *__restrict__ rec, LrSum *__restrict__ sh)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v < s) { const uint4 r = rec2[v]; sh[v] = make_uint2(r.y, r.x == rec[v].x ? LR_NIL : r.x); }
}

__global__ __launch_bounds__(256) void lr_jump_sum_kernel(u64 s, const LrSum *__restrict__ in, LrSum *__restrict__ out)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const LrSum a = in[v];
    if (a.y == LR_NIL) out[v] = a;
    else { const LrSum b = in[a.y]; out[v] = make_uint2(a.x + b.x, b.y); }
}

// One record per LF cycle, for the ordering by smallest element (unbwts.c:62-77).  leader = level-2 leader entry, or
// LR_NIL for a cycle without a splitter (resolved by tiny_cycle_scan_kernel).
struct CycleRec { u32 leader; u32 minelem; u32 len; u32 pad; };

// dist[v] = elements between the leader's first element and v's; the entry whose stretch holds the cycle's smallest
// element publishes that element's distance; leaders append a cycle record
__global__ __launch_bounds__(256) void lr_finish_kernel(u64 s, const LrMin *__restrict__ rec, const LrSum *__restrict__ sh,
                                                        const uint4 *__restrict__ rec2, u32 *__restrict__ dist,
                                                        u32 *__restrict__ min_dist /* by leader */, CycleRec *__restrict__ recs,
                                                        unsigned long long *__restrict__ nrec)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const LrMin r = rec[v];
    const u32 l = r.x;
    const u32 L = sh[l].x;
    const u32 d = L - sh[v].x;
    dist[v] = d;
    const uint4 r2 = rec2[v];
    if (r2.z == r.y) min_dist[l] = d + r2.w;
    if (l == (u32)v) {
        const unsigned long long at = atomicAdd(nrec, 1ull);
        CycleRec c; c.leader = l; c.minelem = r.y; c.len = L; c.pad = 0;
        recs[at] = c;
    }
}

// level-2 entry -> distance of its first element from the cycle's smallest element, the cycle's length and end
__global__ __launch_bounds__(256) void lr_place2_kernel(u64 s, const LrMin *__restrict__ rec, const LrSum *__restrict__ sh,
                                                        const u32 *__restrict__ dist, const u32 *__restrict__ min_dist,
                                                        const u32 *__restrict__ end_by_leader, uint4 *__restrict__ place2)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const u32 l = rec[v].x;
    const u32 L = sh[l].x;                       // a cycle of 2^32 elements reads as 0: the arithmetic below is mod 2^32
    const u32 dm = min_dist[l];
    const u32 d = dist[v];
    const u32 t = d >= dm ? d - dm : d + L - dm;
    place2[v] = make_uint4(t, L, end_by_leader[l], 0u);
}

// hands the positions down to the nodes: opos = text position of the node's first symbol, wrap_at = symbols until the
// walk passes the cycle's smallest element (the text position wraps to the cycle's end there), cyc_len
__global__ __launch_bounds__(256) void lr2_distribute_kernel(const uint4 *__restrict__ noderec, u64 s2, u64 s2all, const u32 *__restrict__ U,
                                                             const uint4 *__restrict__ place2, u32 *__restrict__ opos,
                                                             u32 *__restrict__ wrap_at, u32 *__restrict__ cyc_len)
{
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    if (w >= s2all) return;
    const uint4 pl = place2[w];
    const u64 L = pl.y ? (u64)pl.y : 0x100000000ull;
    u64 t = pl.x;
    if (w >= s2) {
        const u32 v = U[w - s2];
        opos[v] = pl.z - (u32)t; wrap_at[v] = (u32)(L - t); cyc_len[v] = (u32)L;
        return;
    }
    u32 v = (u32)(w * L2_H);
    do {
        const uint4 r = noderec[v];
        opos[v] = pl.z - (u32)t; wrap_at[v] = (u32)(L - t); cyc_len[v] = (u32)L;
        t += r.y;
        if (t >= L) t -= L;
        v = r.x;
    } while (v % L2_H != 0);
}

// ------------------------------------------------------------------------------------
// MARK_MOMENTS: the unreached elements from the ranges' counts, sums and sums of squares
// ------------------------------------------------------------------------------------
// missing = size - count; one missing index: its offset is (sum of all offsets) - (sum seen); two: their sum A and the sum of their
// squares B give (o1 - o2)^2 = 2 B - A^2.  More: the class stays OPEN.
// Real text leaves a few hundred unreached elements (the cycles of its short Lyndon factors), and with 2^10 classes a few dozen
// classes miss three or more: what the arithmetic cannot name, the CYCLES can -- an unreached element's whole cycle is unreached, so
// every element found by arithmetic walks its cycle and the smallest such element of a cycle adds the cycle's members that sit in
// open classes to the list and to their classes' moments; then the open classes are looked at again (fewer missing now), up to
// MOM_PASSES times.  Classes still open after that go onto the list for moments_chase_kernel.  One workgroup: the classes are few.
// cstat[class]: MOM_OPEN, 0 = nothing missing, p = named by arithmetic in pass p.
// counters: [1] unreached elements (as for the other marks), [10] listed classes, [11] the arithmetic did not come out / no room (fall back to the log)
#define MOM_OPEN 0xffffffffu
#define MOM_PASSES 3
__global__ __launch_bounds__(1024) void moments_resolve_kernel(unsigned long long *__restrict__ mom, u32 *__restrict__ cstat, u64 n, int shift /* log2 of the class count */,
                                                               const u32 *__restrict__ LF, u32 *__restrict__ uidx, u32 *__restrict__ ulf, u64 ucap,
                                                               u32 *__restrict__ def_list, unsigned long long *__restrict__ counters, u32 cap)
{
    __shared__ unsigned long long s_from, s_to, s_listed;
    const u64 classes = 1ull << shift;
    const u32 cmask = (u32)classes - 1u;
    for (u64 b = threadIdx.x; b < classes; b += 1024) cstat[b] = MOM_OPEN;
    __threadfence(); __syncthreads();
    for (u32 pass = 1; pass <= MOM_PASSES + 1; pass++) {
        const bool arith = pass <= MOM_PASSES;              // the last look only sorts the classes into complete and listed
        if (threadIdx.x == 0) { s_from = counters[1]; counters[10] = 0; }
        __threadfence(); __syncthreads();
        for (u64 b = threadIdx.x; b < classes && b < n; b += 1024) {
            if (cstat[b] != MOM_OPEN) continue;
            const u64 size = (n - b + classes - 1) >> shift;                 // indices x < n with x mod classes = b: x = o * classes + b, o < size
            const u64 cnt = mom[b];
            if (cnt > size) { atomicAdd(&counters[11], 1ull); continue; }
            const u64 d = size - cnt;
            if (d == 0) { cstat[b] = 0; continue; }
            if (d > 2 || !arith) { const unsigned long long at = atomicAdd(&counters[10], 1ull); def_list[at] = (u32)b; continue; }
            // sums over all offsets 0 .. size - 1 (mod 2^64: the differences below are small and come out exact)
            const u64 sall = size * (size - 1) / 2;
            u64 f[3] = {size - 1, size, 2 * size - 1};                        // (size-1) size (2 size - 1) / 6, dividing before the products overflow
            { int two = 0, three = 0; for (int i = 0; i < 3; i++) { if (!two && f[i] % 2 == 0) { f[i] /= 2; two = 1; } } for (int i = 0; i < 3; i++) { if (!three && f[i] % 3 == 0) { f[i] /= 3; three = 1; } } }
            const u64 qall = f[0] * f[1] * f[2];
            const u64 A = sall - mom[classes + b], B = qall - mom[2 * classes + b];
            if (d == 1) {
                if (A >= size || A * A != B) { atomicAdd(&counters[11], 1ull); continue; }
                const unsigned long long at = atomicAdd(&counters[1], 1ull);
                if (at < ucap) { const u32 x = (u32)((A << shift) | b); uidx[at] = x; ulf[at] = LF[x]; } else atomicAdd(&counters[11], 1ull);
            } else {
                const u64 D = 2 * B - A * A;                                     // (o1 - o2)^2
                u64 r = (u64)sqrt((double)D);
                while (r * r > D) r--;
                while ((r + 1) * (r + 1) <= D) r++;
                const u64 o1 = (A - r) / 2, o2 = (A + r) / 2;
                if (r * r != D || r == 0 || ((A - r) & 1) || o2 >= size || o1 * o1 + o2 * o2 != B) { atomicAdd(&counters[11], 1ull); continue; }
                const unsigned long long at = atomicAdd(&counters[1], 2ull);
                if (at + 1 < ucap) {
                    const u32 x1 = (u32)((o1 << shift) | b), x2 = (u32)((o2 << shift) | b);
                    uidx[at] = x1; ulf[at] = LF[x1]; uidx[at + 1] = x2; ulf[at + 1] = LF[x2];
                } else atomicAdd(&counters[11], 1ull);
            }
            cstat[b] = pass;
        }
        __threadfence(); __syncthreads();
        if (threadIdx.x == 0) { s_to = counters[1]; s_listed = counters[10]; }
        __syncthreads();
        if (!arith || s_listed == 0 || counters[11]) break;              // (the same values for every thread: one decision)
        // the cycles of what this pass named
        const u64 from = s_from, to = s_to < ucap ? s_to : ucap;
        for (u64 i = from + threadIdx.x; i < to; i += 1024) {
            const u32 x = uidx[i];
            bool leader = true, open_seen = false;
            u32 steps = 0;
            for (u32 y = ulf[i]; y != x; y = LF[y]) {
                const u32 st = cstat[y & cmask];
                if (st == pass && y < x) { leader = false; break; }
                if (st == MOM_OPEN) open_seen = true;
                if (++steps > cap) { atomicAdd(&counters[11], 1ull); leader = false; break; }
            }
            if (!leader || !open_seen) continue;
            for (u32 y = ulf[i]; y != x; y = LF[y]) {
                const u32 cls = y & cmask;
                if (cstat[cls] != MOM_OPEN) continue;
                const unsigned long long at = atomicAdd(&counters[1], 1ull);
                if (at < ucap) { uidx[at] = y; ulf[at] = LF[y]; } else atomicAdd(&counters[11], 1ull);
                const unsigned long long o = y >> shift;
                atomicAdd(&mom[cls], 1ull); atomicAdd(&mom[classes + cls], o); atomicAdd(&mom[2 * classes + cls], o * o);
            }
        }
        __threadfence(); __syncthreads();
    }
}
// (own launch, after the one above: all ranges are listed) more elements to search than the budget allows: fall back to the log instead
__global__ void moments_budget_kernel(unsigned long long *__restrict__ counters, u64 per_class, u64 budget)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && counters[10] * per_class > budget) { counters[11] += 1; counters[10] = 0; }
}
// every element of the listed ranges follows LF until it stands on a splitter (a walk came through it: reached) or on itself
// (its cycle holds no splitter: unreached).  `cap` steps without either: counters[11] (the caller falls back to the log).
__global__ __launch_bounds__(256) void moments_chase_kernel(const u32 *__restrict__ def_list, const unsigned long long *__restrict__ counters_in, u64 n, int shift, int g,
                                                            const u32 *__restrict__ LF, u32 cap, u32 *__restrict__ uidx, u32 *__restrict__ ulf, u64 ucap,
                                                            unsigned long long *__restrict__ counters, const u32 *__restrict__ cstat)
{
    const u32 cmask = (1u << shift) - 1u;
    const u64 classes = counters_in[10];
    const u64 members = (n + (1ull << shift) - 1) >> shift;                  // quotients a class may hold
    const u64 per = (members + 255) / 256;                                   // 256-element pieces per class
    const u32 gmask = (1u << g) - 1u;
    for (u64 w = blockIdx.x; w < classes * per; w += gridDim.x) {
        const u64 x0 = (((w % per) * 256 + threadIdx.x) << shift) | (u64)def_list[w / per];
        bool un = false;
        if (x0 < n) {
            if ((x0 & gmask) != 0) {                                         // a splitter is where a walk starts: reached
                // (a cycle that holds an element of a class no longer open is on the list already: moments_resolve_kernel walked it)
                u32 y = LF[x0], steps = 0;
                bool listed = false;
                for (;;) {
                    if (y == (u32)x0) { un = !listed; break; }
                    if ((y & gmask) == 0) break;
                    if (cstat[y & cmask] != MOM_OPEN) listed = true;
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
            if (un) { const u64 at = bse + (u64)__popcll(m & lanemask_lt()); if (at < ucap) { uidx[at] = (u32)x0; ulf[at] = LF[x0]; } }
        }
    }
}

// ------------------------------------------------------------------------------------
// cycles without a splitter (elements no walk reached)
// ------------------------------------------------------------------------------------
// Natural inputs leave a few dozen such elements (tiny Lyndon factors); a constant or sorted input leaves nearly all n
// (LF is close to the identity).  Everything here is sized by their number and stays on the device.
// An unreached element follows its own cycle until it meets a smaller element (not the cycle's minimum: done) or
// returns to itself (it is the minimum: it appends the cycle's record).  `cap` bounds the steps of one lane.
__global__ __launch_bounds__(256) void tiny_cycle_scan_kernel(const u32 *__restrict__ uidx, const u32 *__restrict__ ulf, u64 nu,
                                                              const u32 *__restrict__ LF, u32 cap, uint2 *__restrict__ tiny /* x smallest element, y length */,
                                                              unsigned long long *__restrict__ count, unsigned long long *__restrict__ overflow)
{
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    bool ismin = false;
    u32 x = 0, len = 1;
    if (q < nu) {
        x = uidx[q];
        u32 y = ulf[q];
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
    if (ismin) tiny[b + (u64)__popcll(m & lanemask_lt())] = make_uint2(x, len);
}

// ---- cycle order: by smallest element, the cycle holding index 0 ends the text (unbwts.c:62-77) ------------------------
// record i of the combined list: a cycle without a splitter (i < kt) or a cycle of the reduced list
struct CycleList {
    const uint2 *tiny; u64 kt; const CycleRec *recs; u64 kc;
    __device__ __forceinline__ u32 minelem(u64 i) const { return i < kt ? tiny[i].x : recs[i - kt].minelem; }
    __device__ __forceinline__ u32 len(u64 i) const { return i < kt ? tiny[i].y : recs[i - kt].len; }
};
__global__ __launch_bounds__(256) void cycle_keys_kernel(CycleList cl, u64 *__restrict__ keys, u32 *__restrict__ vals)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < cl.kt + cl.kc) { keys[i] = cl.minelem(i); vals[i] = (u32)i; }
}
struct CycleLenIn {
    CycleList cl; const u32 *order;
    __device__ __forceinline__ u32 operator()(u64 j) const { return cl.len(order[j]); }
};
struct CycleEndOut {
    CycleList cl; const u32 *order; u32 last; u32 *end_by_leader; u32 *end_of_tiny; u64 *total;
    __device__ __forceinline__ void operator()(u64 j, u32 used) const
    {
        const u32 i = order[j];
        const u32 end = last - used;                    // n - 1 - (elements of the cycles ordered before this one)
        if (i < cl.kt) end_of_tiny[i] = end;
        else end_by_leader[cl.recs[i - cl.kt].leader] = end;
        if (j + 1 == cl.kt + cl.kc) *total = (u64)used + cl.len(i);     // == n (mod 2^32 for n = 2^32)
    }
};

// out[end - t] = B[LF^t(min)] for a cycle without a splitter: unbwts.c:73-82, one lane per cycle
__global__ __launch_bounds__(256) void tiny_place_kernel(const uint2 *__restrict__ tiny, u64 kt, const u32 *__restrict__ end_of_tiny,
                                                         const u32 *__restrict__ LF, const u64 *__restrict__ Cg, u8 *__restrict__ out)
{
    __shared__ u64 Ctab[257];
    for (int i = threadIdx.x; i < 257; i += 256) Ctab[i] = Cg[i];
    __syncthreads();
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= kt) return;
    const uint2 c = tiny[i];
    u32 x = c.x, pos = end_of_tiny[i];
    for (u32 t = 0; t < c.y; t++) {
        const u32 y = LF[x];
        out[pos--] = (u8)symbol_of(Ctab, y);
        x = y;
    }
}

// ------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------
static int splitter_log2(const bwts_ctx *ctx, u64 n)
{
    int bl = 0; for (u64 x = n; x; x >>= 1) bl++;
    int g = bl - 24;
    if (g < 4) g = 4;
    if (g > 8) g = 8;
    const char *env = bwts_knob(ctx, "BWTS_SPLIT_LOG2");
    if (env) { int v = atoi(env); if (v >= 0 && v <= 20) g = v; }
    return g;
}

#define UNV_CAP0 (1ull << 20)     // room for unreached elements before their number is known

static size_t inverse_node_bytes(u64 node_cap, u32 slot)
{
    const u64 l2cap = node_cap / L2_H + 2 + node_cap;        // worst case: no node is reached by a level-2 walk
    return align_up(node_cap * 16, 256) + 5 * align_up(node_cap * 4, 256) + align_up(node_cap, 256) + align_up(node_cap * slot, 256) +
           4 * align_up(l2cap * 16, 256) + 2 * align_up(l2cap * 8, 256) + 3 * align_up(l2cap * 4, 256) + align_up(l2cap * sizeof(CycleRec), 256);
}

size_t inverse_arena_bytes(u64 n)
{
    const u64 s = (n >> 4) + 2;   // upper bound on splitters (g >= 4)
    return align_up(n * 4, 256) + radix_tile_hist_bytes(n) + scan_temp_bytes(n) + inverse_node_bytes(s + s / 8 + 1024, 64) + (1 << 16);
}

static int grid1(u64 m) { return (int)((m + 255) / 256); }

#include "wide_inverse.h"         // the 64-bit form; its node-ranking kernels also serve the unit-node ranking below

// wi_finish_kernel for the main path: the cycles of the unit-node ranking go straight into the record form of the cycles
// without a splitter (smallest element, length) plus their leader
__global__ __launch_bounds__(256) void unit_finish_kernel(u64 s, const WiMin *__restrict__ rec, const WiSum *__restrict__ sh, const WiNode *__restrict__ nodes,
                                                          u64 *__restrict__ dist, u64 *__restrict__ min_dist, uint2 *__restrict__ tiny, u32 *__restrict__ leader,
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
        tiny[at] = make_uint2((u32)r.mn, (u32)L);
        leader[at] = r.leader;
    }
}
__global__ __launch_bounds__(256) void unit_ends_kernel(const u32 *__restrict__ leader, const u32 *__restrict__ end_of_tiny, u64 m, u32 *__restrict__ end_by_leader)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < m) end_by_leader[leader[i]] = end_of_tiny[i];
}

// One attempt with splitter spacing 2^g.  *retry is set when the node pool overflows (adversarial LF) or the unreached
// elements are too many for the unit-node ranking; the caller then repeats with g = 0 (every element a splitter).
static int inverse_attempt(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out, int g, int mark, bool *retry, bool *ambiguous, bool *need_log = nullptr)
{
    *retry = false;
    *ambiguous = false;
    if (need_log) *need_log = false;
    const bool moments = mark == MARK_MOMENTS;
    const int mom_shift = n > (1ull << 30) ? MOM_LOG2_LARGE : MOM_LOG2_SMALL;       // log2 of the number of residue classes
    const u64 mom_classes = 1ull << mom_shift;
    const u64 G = 1ull << g;
    const u64 s = (n + G - 1) / G;
    const u64 tiles = (n + LF_TILE - 1) / LF_TILE;

    // a segment longer than `slot` steps is cut into virtual nodes; room for s/8 of them (natural data needs ~2 %)
    const u32 slot = (u32)(4 * G < 16 ? 16 : 4 * G);
    const u64 node_cap = g == 0 ? s : s + s / 8 + 1024;
    const u64 l2cap = node_cap / L2_H + 2 + node_cap;
    u64 walker_cap = 524288;
    if (const char *e = bwts_knob(ctx, "BWTS_WALKERS")) { const long v = atol(e); if (v >= 256 && v <= (1 << 22)) walker_cap = (u64)v; }
    const u64 walkers = s < walker_cap ? s : walker_cap;
    const unsigned wblocks = (unsigned)((walkers + 255) / 256);
    const u64 log_chunks = n / (IDX_CHUNK - 64) + (u64)wblocks * 4 + 2;     // a closed chunk wastes < 64 entries; every wave may leave one open
    BWTS_TRY(arena_reserve(ctx, align_up(n * 4, 256) + radix_tile_hist_bytes(n) + scan_temp_bytes(n) + inverse_node_bytes(node_cap, slot) +
                                    align_up(n, 256) +
                                    (mark == MARK_LOG ? align_up(log_chunks * IDX_CHUNK * 4, 256) + align_up(log_chunks * 4, 256) + align_up(n * 4 + (4ull << IDX_RANGE_LOG2), 256) + (1 << 20) : 0) +
                                    (1 << 18)));
    u32 *LF = arena_array<u32>(ctx, n);
    unsigned long long *mom = moments ? (unsigned long long *)arena_array<u64>(ctx, 3 * MOM_MAX_BUCKETS) : nullptr;
    unsigned long long *mom_work = moments ? (unsigned long long *)arena_array<u64>(ctx, 3 * MOM_MAX_BUCKETS) : nullptr;   // (moments_resolve_kernel adds to its copy)
    u32 *def_list = moments ? arena_array<u32>(ctx, MOM_MAX_BUCKETS) : nullptr, *cstat = moments ? arena_array<u32>(ctx, MOM_MAX_BUCKETS) : nullptr;
    if (moments && (!mom || !mom_work || !def_list || !cstat)) return BWTS_E_NOMEM;
    u32 *tile_hist = (u32 *)arena_alloc(ctx, radix_tile_hist_bytes(n));
    void *scan_temp = arena_alloc(ctx, scan_temp_bytes(n));
    uint4 *noderec = arena_array<uint4>(ctx, node_cap);
    u32 *d_opos = arena_array<u32>(ctx, node_cap), *d_wrap = arena_array<u32>(ctx, node_cap), *d_clen = arena_array<u32>(ctx, node_cap);
    u32 *id2of = arena_array<u32>(ctx, node_cap), *U = arena_array<u32>(ctx, node_cap);
    u8 *visited2 = arena_array<u8>(ctx, node_cap);
    u8 *seg = arena_array<u8>(ctx, node_cap * slot);
    uint4 *rec2 = arena_array<uint4>(ctx, l2cap), *place2 = arena_array<uint4>(ctx, l2cap);
    LrMin *lrmin[2] = {arena_array<LrMin>(ctx, l2cap), arena_array<LrMin>(ctx, l2cap)};
    LrSum *lrsum[2] = {arena_array<LrSum>(ctx, l2cap), arena_array<LrSum>(ctx, l2cap)};
    u32 *dist = arena_array<u32>(ctx, l2cap), *min_dist = arena_array<u32>(ctx, l2cap), *end_by_leader = arena_array<u32>(ctx, l2cap);
    CycleRec *d_recs2 = (CycleRec *)arena_alloc(ctx, l2cap * sizeof(CycleRec));      // cycles of the reduced list (copied next to the others later)
    const bool bytemark = mark == MARK_BYTEMAP;
    u8 *marks = bytemark ? arena_array<u8>(ctx, n) : nullptr;
    const u32 nbuckets = (u32)((n + (1ull << IDX_RANGE_LOG2) - 1) >> IDX_RANGE_LOG2);
    u32 *idxlog = mark == MARK_LOG ? arena_array<u32>(ctx, log_chunks * IDX_CHUNK) : nullptr;
    u32 *chunk_fill = mark == MARK_LOG ? arena_array<u32>(ctx, log_chunks) : nullptr;
    u32 *bucket_data = mark == MARK_LOG ? arena_array<u32>(ctx, (u64)nbuckets << IDX_RANGE_LOG2) : nullptr;
    u32 *bucket_fill = mark == MARK_LOG ? arena_array<u32>(ctx, (u64)IDX_MAX_BUCKETS * IDX_FILL_STRIDE) : nullptr;
    u32 *bucket_seen = mark == MARK_LOG ? arena_array<u32>(ctx, 2 * IDX_MAX_BUCKETS) : nullptr;   // counts, then deficit flags
    u32 *deficient = bucket_seen ? bucket_seen + IDX_MAX_BUCKETS : nullptr;
    if (!LF || !tile_hist || !scan_temp || !noderec || !d_opos || !d_wrap || !d_clen || !id2of || !U || !visited2 || !seg || !rec2 || !place2 ||
        !lrmin[0] || !lrmin[1] || !lrsum[0] || !lrsum[1] || !dist || !min_dist || !end_by_leader || !d_recs2 || (bytemark && !marks) ||
        (mark == MARK_LOG && (!idxlog || !chunk_fill || !bucket_data || !bucket_fill || !bucket_seen)))
        return BWTS_E_NOMEM;
    if (bytemark) HIPC(hipMemsetAsync(marks, 0, n, ctx->stream));

    u64 *dC = ctx->d_small + 1024;          // symbol boundaries C[0..256] (unbwts.c:38-43); filled below

    // stable LF map (unbwts.c:50-52)
    {
        SpanGuard sg(ctx, BWTS_K_LF_BUILD, n, 5 * n);
        lf_hist_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in, n, tile_hist);
        BWTS_TRY(radix_column_scan(ctx, tile_hist, tiles, scan_temp));
        // the scanned table's first row is C itself: no separate histogram sweep, no host round trip before the walk
        ctab_from_tiles_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(tile_hist, n, d_in, dC);
        lf_rank_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in, n, tile_hist, LF);
        HIPC(hipGetLastError());
    }

    // the walk: marks, segment symbols, reduced list
    // counters: [0] ticket, [1] unreached elements, [2] cycles of the reduced list, [3] virtual nodes, [4] overflow, [5] log chunks,
    //           [6] unreached nodes, [7] cycles without a splitter, [13] sum of all cycle lengths ([8..12]: WALK_PROFILE stamps)
    unsigned long long *ticket = (unsigned long long *)(ctx->d_small + SMI_COUNTERS);
    HIPC(hipMemsetAsync(ticket, 0, 16 * sizeof(u64), ctx->stream));
#ifdef WALK_PROFILE
    HIPC(hipMemsetAsync(ticket + 8, 0xff, 2 * sizeof(u64), ctx->stream));
    HIPC(hipMemsetAsync(ticket + 12, 0xff, 1 * sizeof(u64), ctx->stream));
#endif
    if (mark == MARK_LOG) {
        HIPC(hipMemsetAsync(chunk_fill, 0, log_chunks * sizeof(u32), ctx->stream));
        HIPC(hipMemsetAsync(bucket_seen, 0, IDX_MAX_BUCKETS * sizeof(u32), ctx->stream));
    }
    if (moments) HIPC(hipMemsetAsync(mom, 0, 3 * mom_classes * sizeof(u64), ctx->stream));
    {
        SpanGuard sg(ctx, BWTS_K_WALK, n, 6 * n);
        if (moments) {
            BWTS_TRY(ensure_dyn_lds(ctx, (const void *)walk_record_kernel<MARK_MOMENTS>, (size_t)MOM_MAX_BUCKETS * 20));
            walk_record_kernel<MARK_MOMENTS><<<dim3(wblocks), dim3(256), (size_t)mom_classes * 20, ctx->stream>>>(LF, marks, idxlog, s, node_cap, g, slot, dC, seg, noderec,
                                                                                          ticket, ticket + 3, ticket + 4, ticket + 5, chunk_fill, log_chunks, nbuckets, bucket_seen,
                                                                                          mom_shift, mom);
        } else if (mark == MARK_BYTEMAP)
            walk_record_kernel<MARK_BYTEMAP><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, marks, idxlog, s, node_cap, g, slot, dC, seg, noderec,
                                                                                          ticket, ticket + 3, ticket + 4, ticket + 5, chunk_fill, log_chunks, nbuckets, bucket_seen);
        else if (mark == MARK_SENTINEL)
            walk_record_kernel<MARK_SENTINEL><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, marks, idxlog, s, node_cap, g, slot, dC, seg, noderec,
                                                                                           ticket, ticket + 3, ticket + 4, ticket + 5, chunk_fill, log_chunks, nbuckets, bucket_seen);
        else
            walk_record_kernel<MARK_LOG><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, marks, idxlog, s, node_cap, g, slot, dC, seg, noderec,
                                                                                      ticket, ticket + 3, ticket + 4, ticket + 5, chunk_fill, log_chunks, nbuckets, bucket_seen);
        HIPC(hipGetLastError());
    }
    // virtual nodes join the reduced list: its size is only known now
    BWTS_TRY(read_small(ctx, SMI_COUNTERS, 16));
#ifdef WALK_PROFILE
    {
        const u64 *pf = ctx->h_small + SMI_COUNTERS + 8;
        fprintf(stderr, "walk profile (ms from first wave start): first-exhausted %.2f  last-exhausted %.2f  first-wave-end %.2f  last-wave-end %.2f\n",
                (pf[1] - pf[0]) / 1e5, (pf[3] - pf[0]) / 1e5, (pf[4] - pf[0]) / 1e5, (pf[2] - pf[0]) / 1e5);
    }
#endif
    if (ctx->h_small[SMI_COUNTERS + 4]) { *retry = true; return BWTS_OK; }   // node pool exhausted (adversarial LF): plain pointer jumping
    const u64 s_all = s + ctx->h_small[SMI_COUNTERS + 3];
    const u64 s2 = (s_all + L2_H - 1) / L2_H;

    // elements in cycles without a splitter: collected into lists sized by their number (a second pass if the first room is too small)
    size_t ucap = ctx->unv_hint > UNV_CAP0 ? ctx->unv_hint : UNV_CAP0;
    if (ucap > n) ucap = (size_t)n;
    u32 *uidx = nullptr, *ulf = nullptr, *end_of_tiny = nullptr;
    uint2 *tiny = nullptr;
    auto lay_out_lists = [&](size_t cap) -> int {
        // uidx, ulf: cap entries; records of the cycles without a splitter and their ends: at most cap
        char *ub = nullptr;
        const size_t e4 = align_up(cap * 4, 256), e8 = align_up(cap * 8, 256);
        BWTS_TRY(aux_reserve_slot(ctx, 0, 3 * e4 + e8, &ub));
        uidx = (u32 *)ub; ulf = (u32 *)(ub + e4);
        tiny = (uint2 *)(ub + 2 * e4);
        end_of_tiny = (u32 *)(ub + 2 * e4 + e8);
        return BWTS_OK;
    };
    BWTS_TRY(lay_out_lists(ucap));
    auto collect_unreached = [&](bool first_time) -> int {
        u64 blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
        if (moments) {
            if (!first_time) HIPC(hipMemsetAsync(ticket + 10, 0, 2 * sizeof(u64), ctx->stream));
            const u64 per_class = (n + mom_classes - 1) >> mom_shift;
            const u64 budget = (4ull << 20) > per_class ? (4ull << 20) : per_class;        // elements the search may look at (at least one class)
            HIPC(hipMemcpyAsync(mom_work, mom, 3 * mom_classes * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
            moments_resolve_kernel<<<dim3(1), dim3(1024), 0, ctx->stream>>>(mom_work, cstat, n, mom_shift, LF, uidx, ulf, ucap, def_list, ticket, 1u << 16);
            moments_budget_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(ticket, per_class, budget);
            moments_chase_kernel<<<dim3(2048), dim3(256), 0, ctx->stream>>>(def_list, ticket, n, mom_shift, g, LF, 1u << 16, uidx, ulf, ucap, ticket, cstat);
        } else if (mark == MARK_LOG) {
            const int bm_bytes = (int)((1u << IDX_RANGE_LOG2) / 8);
            if (first_time) {
                HIPC(hipMemsetAsync(bucket_fill, 0, (size_t)nbuckets * IDX_FILL_STRIDE * sizeof(u32), ctx->stream));
                BWTS_TRY(ensure_dyn_lds(ctx, (const void *)unvisited_from_buckets_kernel, (size_t)bm_bytes));
                BWTS_TRY(ensure_dyn_lds(ctx, (const void *)bucket_indices_kernel, bucket_indices_lds_bytes(IDX_MAX_BUCKETS)));
                bucket_deficit_kernel<<<dim3((nbuckets + 255) / 256), dim3(256), 0, ctx->stream>>>(bucket_seen, nbuckets, n, deficient);
                bucket_indices_kernel<<<dim3((unsigned)log_chunks), dim3(IDX_THREADS), bucket_indices_lds_bytes(nbuckets), ctx->stream>>>(
                    idxlog, chunk_fill, nbuckets, deficient, bucket_fill, bucket_data);
            }
            unvisited_from_buckets_kernel<<<dim3(nbuckets), dim3(1024), bm_bytes, ctx->stream>>>(deficient, bucket_fill, bucket_data, LF, n, uidx, ulf, ucap,
                                                                                                ticket + 1);
        } else if (mark == MARK_BYTEMAP)
            collect_unvisited_kernel<MARK_BYTEMAP><<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(LF, marks, n, uidx, ulf, ucap, ticket + 1);
        else
            collect_unvisited_kernel<MARK_SENTINEL><<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(LF, marks, n, uidx, ulf, ucap, ticket + 1);
        HIPC(hipGetLastError());
        return BWTS_OK;
    };
    {
        SpanGuard sg(ctx, BWTS_K_OTHER, n, 4 * n);
        BWTS_TRY(collect_unreached(true));
    }
    // level-2 walk over the node list, and the nodes it does not reach
    {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, s_all, 32 * s_all);
        HIPC(hipMemsetAsync(visited2, 0, s_all, ctx->stream));
        lr2_walk_kernel<<<dim3(grid1(s2)), dim3(256), 0, ctx->stream>>>(noderec, s2, visited2, rec2);
        lr2_collect_kernel<<<dim3(grid1(s_all)), dim3(256), 0, ctx->stream>>>(visited2, s_all, s2, U, id2of, ticket + 6);
        HIPC(hipGetLastError());
    }
    BWTS_TRY(read_small(ctx, SMI_COUNTERS, 16));
    const u64 nu = ctx->h_small[SMI_COUNTERS + 1];
    const u64 nu2 = ctx->h_small[SMI_COUNTERS + 6];
    const bool inv_trace = [ctx] { const char *e = bwts_knob(ctx, "BWTS_INV_TRACE"); return e && atoi(e) == 1; }();
    if (inv_trace && moments)
        fprintf(stderr, "[inverse] moments: shift %d, unreached found %llu, ranges searched %llu, fallback flag %llu\n", mom_shift,
                (unsigned long long)ctx->h_small[SMI_COUNTERS + 1], (unsigned long long)ctx->h_small[SMI_COUNTERS + 10], (unsigned long long)ctx->h_small[SMI_COUNTERS + 11]);
    if (moments && ctx->h_small[SMI_COUNTERS + 11]) {          // the ranges' moments do not name the unreached elements: the index log does
        if (need_log) *need_log = true;
        return BWTS_OK;
    }
    ctx->tm.unvisited = nu;
    if (nu > n || nu2 > s_all) return BWTS_E_INTERNAL;
    ctx->unv_hint = (size_t)nu;
    if (nu > ucap) {
        ucap = (size_t)nu;
        BWTS_TRY(lay_out_lists(ucap));
        HIPC(hipMemsetAsync(ticket + 1, 0, sizeof(u64), ctx->stream));
        SpanGuard sg(ctx, BWTS_K_OTHER, n, 4 * n);
        BWTS_TRY(collect_unreached(false));
    }
    const u64 s2all = s2 + nu2;

    // level-2 list ranking by pointer jumping; cycle records of the reduced list; cycles without a splitter
    const int R = [&] { int b = 0; for (u64 x = s2all; x; x >>= 1) b++; return b; }();   // 2^R > s2all >= any cycle's entry count
    int cur = 0, sc = 0;
    {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, s2all, 0);
        const int gb = grid1(s2all);
        if (nu2) lr2_fill_kernel<<<dim3(grid1(nu2)), dim3(256), 0, ctx->stream>>>(U, nu2, s2, noderec, id2of, rec2);
        lr_init_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s2all, rec2, lrmin[0]);
        for (int r = 0; r < R; r++, cur ^= 1)
            lr_jump_min_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s2all, lrmin[cur], lrmin[cur ^ 1]);
        lr_cut_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s2all, rec2, lrmin[cur], lrsum[0]);
        for (int r = 0; r < R; r++, sc ^= 1)
            lr_jump_sum_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s2all, lrsum[sc], lrsum[sc ^ 1]);
        lr_finish_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s2all, lrmin[cur], lrsum[sc], rec2, dist, min_dist, d_recs2, ticket + 2);
        HIPC(hipGetLastError());
    }
    if (nu) {
        SpanGuard sg(ctx, BWTS_K_OTHER, nu, 8 * nu);
        // one lane follows at most `cap` elements: bounds the work of an adversarial LF (many long cycles that dodge every splitter)
        u64 cap = (1ull << 36) / nu;
        if (cap > 64 * G) cap = 64 * G;
        if (cap < 4 * G) cap = 4 * G;
        tiny_cycle_scan_kernel<<<dim3(grid1(nu)), dim3(256), 0, ctx->stream>>>(uidx, ulf, nu, LF, (u32)(cap > 0xfffffff0ull ? 0xfffffff0ull : cap),
                                                                              tiny, ticket + 7, ticket + 4);
        HIPC(hipGetLastError());
    }
    BWTS_TRY(read_small(ctx, SMI_COUNTERS, 16));
    const u64 kc = ctx->h_small[SMI_COUNTERS + 2];
    u64 kt = ctx->h_small[SMI_COUNTERS + 7];
    if (kc == 0 || kc > s2all || kt > nu) return BWTS_E_INTERNAL;
    // A cycle without a splitter too long for one lane (sorted or periodic data: 1^b 0^c with n = 2^k, c = 2 * odd has a cycle of
    // n / 2 odd elements): every unreached element becomes a node of one symbol and the pointer-jumping kernels of the 64-bit
    // form rank that list -- memory and work by the number of unreached elements, not by n (wide_inverse.h).
    const bool unit_rank = ctx->h_small[SMI_COUNTERS + 4] != 0;
    ScopedDeviceBlock ub(ctx);
    WiMin *umin[2] = {nullptr, nullptr};
    WiSum *usum[2] = {nullptr, nullptr};
    u64 *udist = nullptr, *umind = nullptr;
    u32 *uend = nullptr, *uleader = nullptr;
    int ucur = 0, usc = 0;
    if (unit_rank) {
        if (nu >= 0x7ffffff0ull) { *retry = true; return BWTS_OK; }
        SpanGuard sg(ctx, BWTS_K_LISTRANK, nu, 0);
        const size_t a24 = align_up(nu * sizeof(WiNode), 256), a16 = align_up(nu * 16, 256), a8 = align_up(nu * 8, 256), a4 = align_up(nu * 4, 256);
        if (ub.take(a24 + 4 * a16 + 2 * a8 + 2 * a4) != BWTS_OK) { *retry = true; return BWTS_OK; }      // 112 bytes per unreached element
        WiNode *unodes = (WiNode *)ub.p;
        umin[0] = (WiMin *)(ub.p + a24); umin[1] = (WiMin *)(ub.p + a24 + a16);
        usum[0] = (WiSum *)(ub.p + a24 + 2 * a16); usum[1] = (WiSum *)(ub.p + a24 + 3 * a16);
        udist = (u64 *)(ub.p + a24 + 4 * a16); umind = (u64 *)(ub.p + a24 + 4 * a16 + a8);
        uend = (u32 *)(ub.p + a24 + 4 * a16 + 2 * a8); uleader = (u32 *)(ub.p + a24 + 4 * a16 + 2 * a8 + a4);
        const int gb = grid1(nu);
        const int R2 = [&] { int b = 0; for (u64 x = nu; x; x >>= 1) b++; return b; }();
        HIPC(hipMemsetAsync(ticket + 9, 0, sizeof(u64), ctx->stream));
        wi_unit_index_kernel<u32><<<dim3(gb), dim3(256), 0, ctx->stream>>>(uidx, nu, LF);        // LF[x] of an unreached x lives on in ulf
        wi_unit_nodes_kernel<u32><<<dim3(gb), dim3(256), 0, ctx->stream>>>(uidx, ulf, nu, LF, unodes);
        wi_init_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, unodes, umin[0]);
        for (int r = 0; r < R2; r++, ucur ^= 1) wi_jump_min_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, umin[ucur], umin[ucur ^ 1]);
        wi_cut_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, unodes, umin[ucur], usum[0]);
        for (int r = 0; r < R2; r++, usc ^= 1) wi_jump_sum_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, usum[usc], usum[usc ^ 1]);
        unit_finish_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(nu, umin[ucur], usum[usc], unodes, udist, umind, tiny, uleader, ticket + 9);
        HIPC(hipGetLastError());
        BWTS_TRY(read_small(ctx, SMI_COUNTERS + 9, 1));
        kt = ctx->h_small[SMI_COUNTERS + 9];              // these cycles take the place of the one-lane scan's
        if (kt == 0 || kt > nu) return BWTS_E_INTERNAL;
    }
    const u64 kall = kc + kt;
    ctx->tm.factors = kall;

    // order the cycles by smallest element on the device: sort (minelem, record), prefix sums of the lengths
    {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, kall, 0);
        const CycleList cl{tiny, kt, d_recs2, kc};
        char *sb = nullptr;
        const size_t k8 = align_up(kall * 8, 256), k4 = align_up(kall * 4, 256);
        BWTS_TRY(aux_reserve_slot(ctx, 1, 2 * k8 + 2 * k4 + radix_tile_hist_bytes(kall) + scan_temp_bytes(kall) + 4096, &sb));
        SortPlan cp;
        cp.keys[0] = (u64 *)sb; cp.keys[1] = (u64 *)(sb + k8);
        cp.vals[0] = (u32 *)(sb + 2 * k8); cp.vals[1] = (u32 *)(sb + 2 * k8 + k4);
        cp.tile_hist = (u32 *)(sb + 2 * k8 + 2 * k4);
        cp.scan_temp = sb + 2 * k8 + 2 * k4 + radix_tile_hist_bytes(kall);
        cycle_keys_kernel<<<dim3(grid1(kall)), dim3(256), 0, ctx->stream>>>(cl, cp.keys[0], cp.vals[0]);
        int res = 0;
        int kbits = 0; for (u64 x = n - 1; x; x >>= 1) kbits++;
        BWTS_TRY(radix_sort_pairs(ctx, cp, kall, kbits < 1 ? 1 : kbits, &res));
        CycleLenIn lin{cl, cp.vals[res]};
        CycleEndOut lout{cl, cp.vals[res], (u32)(n - 1), end_by_leader, end_of_tiny, ctx->d_small + SMI_COUNTERS + 13};
        BWTS_TRY((device_scan<false, u32>(ctx, kall, lin, lout, OpAdd(), 0u, cp.scan_temp)));
        if (unit_rank) unit_ends_kernel<<<dim3(grid1(kt)), dim3(256), 0, ctx->stream>>>(uleader, end_of_tiny, kt, uend);
        lr_place2_kernel<<<dim3(grid1(s2all)), dim3(256), 0, ctx->stream>>>(s2all, lrmin[cur], lrsum[sc], dist, min_dist, end_by_leader, place2);
        lr2_distribute_kernel<<<dim3(grid1(s2all)), dim3(256), 0, ctx->stream>>>(noderec, s2, s2all, U, place2, d_opos, d_wrap, d_clen);
        HIPC(hipGetLastError());
    }

    // the recorded segments go to their places in the text (unbwts.c:73-82)
    {
        SpanGuard sg(ctx, BWTS_K_WALK_EMIT, n, 2 * n);
        int tpn_log2 = g - 4;                           // one 16-symbol chunk per thread at the expected segment length (G)
        if (tpn_log2 < 0) tpn_log2 = 0;
        if (tpn_log2 > 8) tpn_log2 = 8;
        const u64 threads = s_all << tpn_log2;
        place_segments_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream>>>(seg, s_all, slot, tpn_log2, noderec,
                                                                                                      d_opos, d_wrap, d_clen, d_out);
        if (unit_rank)
            wi_unit_place_kernel<u32><<<dim3(grid1(nu)), dim3(256), 0, ctx->stream>>>(nu, umin[ucur], usum[usc], udist, umind, uend, ulf, dC, d_out);
        else if (kt) tiny_place_kernel<<<dim3(grid1(kt)), dim3(256), 0, ctx->stream>>>(tiny, kt, end_of_tiny, LF, dC, d_out);
        HIPC(hipGetLastError());
    }
    BWTS_TRY(read_small(ctx, SMI_COUNTERS + 13, 1));
    if ((u32)ctx->h_small[SMI_COUNTERS + 13] != (u32)n) {
        // n = 2^32 only: the one entry whose value equals LF_VISITED sat in a cycle without a splitter and was taken for visited
        if (mark == MARK_SENTINEL && n == 0x100000000ull) { *ambiguous = true; return BWTS_OK; }
        return BWTS_E_INTERNAL;
    }
    return BWTS_OK;
}

int inverse_device_impl(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out)
{
    // beyond 32-bit indices: the 64-bit form (wide_inverse.h); BWTS_FORCE_WIDE sends every input there (tests)
    const int force_wide = [ctx] { const char *e = bwts_knob(ctx, "BWTS_FORCE_WIDE"); return e ? atoi(e) : 0; }();
    ctx->tm.attempts = 1;
    if (n > 0x80000000ull) {
        // one byte value only: LF is the identity, n cycles of one element, the text is the input (unbwts.c:66-86 walks each of them
        // in one step).  The general route holds ~110 bytes per element of cycles that meet no splitter -- all of them here --, which
        // beyond 2^31 elements no device has; so large inputs are looked at first (eight probes, then the histogram only if they agree).
        bool constant = false;
        BWTS_TRY(constant_input_probe(ctx, d_in, n, &constant));
        if (constant) {
            HIPC(hipMemcpyAsync(d_out, d_in, n, hipMemcpyDefault, ctx->stream));
            ctx->tm.factors = n; ctx->tm.unvisited = 0;
            return BWTS_OK;
        }
    }
    if (n > 0x100000000ull || force_wide) return inverse_wide_impl(ctx, d_in, n, d_out);
    bool retry = false, ambiguous = false;
    // how the unreached elements are found: per-range moments (default; falls back to the index log when too many are missing),
    // the index log, or the two mark forms (BWTS_INV_MARK=log|sentinel|bytemap, BWTS_BYTEMARK=1: tests, and the fallback chain below)
    int mark = MARK_MOMENTS;
    const char *me = bwts_knob(ctx, "BWTS_INV_MARK");
    if (me && !strcmp(me, "log")) mark = MARK_LOG;
    if (me && !strcmp(me, "sentinel")) mark = MARK_SENTINEL;
    if ((me && !strcmp(me, "bytemap")) || bwts_knob(ctx, "BWTS_BYTEMARK")) mark = MARK_BYTEMAP;
    bool need_log = false;
    // Worst case: the walk runs up to five times (moments -> index log -> byte map at n = 2^32 -> every element a splitter, with
    // sentinel and then byte-map marks); natural inputs take one.  bwts_timings.attempts says how many it was.
    ctx->tm.attempts = 1;
    BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, splitter_log2(ctx, n), mark, &retry, &ambiguous, &need_log));
    if (need_log) {             // many unreached elements (low-entropy input): the walk again, this time logging every index it visits
        mark = MARK_LOG;
        ctx->tm.attempts++;
        BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, splitter_log2(ctx, n), mark, &retry, &ambiguous));
    }
    if (ambiguous) {            // sentinel marks only, n = 2^32: 0xffffffff was a real entry of a cycle without a splitter
        mark = MARK_BYTEMAP;
        ctx->tm.attempts++;
        BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, splitter_log2(ctx, n), mark, &retry, &ambiguous));
    }
    if (retry) {
        if (mark == MARK_LOG || mark == MARK_MOMENTS) mark = MARK_SENTINEL;        // adversarial LF: keep the retry on the simplest marks
        ctx->tm.attempts++;
        BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, 0, mark, &retry, &ambiguous));
        if (ambiguous) { ctx->tm.attempts++; BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, 0, MARK_BYTEMAP, &retry, &ambiguous)); }
        if (retry || ambiguous) return BWTS_E_INTERNAL;
    }
    return BWTS_OK;
}
