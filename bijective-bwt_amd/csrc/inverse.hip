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

#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <vector>

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
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int bb = 0; bb < 4; bb++) atomicAdd(&mine[(ws[a] >> (8 * bb)) & 255u], 1u);
        }
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
// a boundary equal to 2^32 (n = 2^32, symbols above the largest one present) reads as a value below its predecessor.
__global__ void ctab_from_tiles_kernel(const u32 *__restrict__ tile_off, u64 n, u64 *__restrict__ C)
{
    if (threadIdx.x != 0) return;
    u64 prev = 0;
    for (int c = 0; c < 256; c++) {
        u64 v = tile_off[c];
        if (v < prev) v += 0x100000000ull;
        C[c] = v;
        prev = v;
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
template <int MARK>
__global__ __launch_bounds__(256) void walk_record_kernel(u32 *__restrict__ LF, u8 *__restrict__ marks, u32 *__restrict__ idxlog, u64 s, u64 node_cap, int g, u32 slot,
                                                          const u64 *__restrict__ Cg, u8 *__restrict__ seg,
                                                          u32 *__restrict__ nxt, u32 *__restrict__ seglen,
                                                          u32 *__restrict__ segmin, u32 *__restrict__ segminoff,
                                                          unsigned long long *__restrict__ ticket,
                                                          unsigned long long *__restrict__ vcount,
                                                          unsigned long long *__restrict__ overflow,
                                                          unsigned long long *__restrict__ chunk_ctr, u32 *__restrict__ chunk_fill, u64 log_chunks,
                                                          u32 nbuckets, u32 *__restrict__ bucket_seen)
{
    __shared__ u64 Ctab[257];
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
    u32 sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0;     // 16 recorded symbols waiting for one 16-byte store
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
                if (id < bend) { have = true; my = id; x = (u32)(my << g); len = 0; mn = x; mnoff = 0; sb0 = sb1 = sb2 = sb3 = 0; }
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
        if (have) {
            const u32 y = LF[x];
            if (MARK == MARK_BYTEMAP) marks[x] = 1;
            else if (MARK == MARK_SENTINEL) LF[x] = LF_VISITED;       // the entry is not needed again
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
                if (len & 15u)                                                                               // slot is a multiple of 16
                    *(uint4 *)(seg + my * slot + (len & ~15u)) = make_uint4(sb0, sb1, sb2, sb3);
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
    if (MARK == MARK_LOG && lend && lane_id() == 0) chunk_fill[lbase / IDX_CHUNK] = (u32)(lcur - lbase);
    if (MARK == MARK_LOG) {
        __syncthreads();                  // every wave leaves the loop (the pool runs dry for all of them)
        for (u32 b = threadIdx.x; b < nbuckets; b += 256) { const u32 c = bseen[b]; if (c) atomicAdd(&bucket_seen[b], c); }
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
    const u64 lo = (u64)b << IDX_RANGE_LOG2;
    for (u32 w = threadIdx.x; w < (1u << IDX_RANGE_LOG2) / 32; w += 1024) {
        u32 zeros = ~bm[w];
        while (zeros) {
            const u32 bit = (u32)__ffs((int)zeros) - 1u;
            zeros &= zeros - 1u;
            const u64 x = lo + (u64)w * 32 + bit;
            if (x < n) {
                const unsigned long long at = atomicAdd(count, 1ull);      // rare: a few thousand on natural inputs
                if (at < cap) { uidx[at] = (u32)x; ulf[at] = LF[x]; }
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

// A round of pointer jumping reads the record of the node it hops to: the fields travel together (16 and 8 bytes), so
// a round costs one random line fill per node instead of three (two).
//   LrMin  x = smallest node id seen (-> leader), y = smallest element seen, z = hop target
//   LrSum  x = segment lengths summed up to the cut, y = hop target (LR_NIL at the cut)
typedef uint4 LrMin;
typedef uint2 LrSum;

__global__ __launch_bounds__(256) void lr_init_kernel(u64 s, const u32 *__restrict__ nxt, const u32 *__restrict__ mn, LrMin *__restrict__ rec)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v < s) rec[v] = make_uint4((u32)v, mn[v], nxt[v], 0u);
}

// after r rounds a node has folded in the 2^r nodes that follow it
__global__ __launch_bounds__(256) void lr_jump_min_kernel(u64 s, const LrMin *__restrict__ in, LrMin *__restrict__ out)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const LrMin a = in[v];
    const LrMin b = in[a.z];
    out[v] = make_uint4(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, b.z, 0u);
}

// cut every cycle in front of its leader, then suffix sums of the segment lengths
__global__ __launch_bounds__(256) void lr_cut_kernel(u64 s, const u32 *__restrict__ nxt, const u32 *__restrict__ len,
                                                     const LrMin *__restrict__ rec, LrSum *__restrict__ sh)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v < s) { const u32 nx = nxt[v]; sh[v] = make_uint2(len[v], nx == rec[v].x ? LR_NIL : nx); }
}

__global__ __launch_bounds__(256) void lr_jump_sum_kernel(u64 s, const LrSum *__restrict__ in, LrSum *__restrict__ out)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const LrSum a = in[v];
    if (a.y == LR_NIL) out[v] = a;
    else { const LrSum b = in[a.y]; out[v] = make_uint2(a.x + b.x, b.y); }
}

struct CycleRec { u32 leader; u32 minelem; u32 len; u32 pad; };

// dist[v] = elements between the leader's splitter and v's splitter; the node whose segment holds the
// cycle's smallest element publishes that element's distance; leaders append a cycle record
__global__ __launch_bounds__(256) void lr_finish_kernel(u64 s, const LrMin *__restrict__ rec, const LrSum *__restrict__ sh,
                                                        const u32 *__restrict__ mn, const u32 *__restrict__ off, u32 *__restrict__ dist,
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
    if (mn[v] == r.y) min_dist[l] = d + off[v];
    if (l == (u32)v) {
        const unsigned long long at = atomicAdd(nrec, 1ull);
        CycleRec c; c.leader = l; c.minelem = r.y; c.len = L; c.pad = 0;
        recs[at] = c;
    }
}

__global__ __launch_bounds__(256) void lr_place_kernel(u64 s, const LrMin *__restrict__ rec, const LrSum *__restrict__ sh,
                                                       const u32 *__restrict__ dist, const u32 *__restrict__ min_dist,
                                                       const u32 *__restrict__ end_by_leader, u32 *__restrict__ opos,
                                                       u32 *__restrict__ wrap_at, u32 *__restrict__ cyc_len)
{
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= s) return;
    const u32 l = rec[v].x;
    const u32 L = sh[l].x;
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
static int inverse_attempt(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out, int g, int mark, bool *retry, bool *ambiguous)
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
    u64 walker_cap = 524288;
    if (const char *e = getenv("BWTS_WALKERS")) { const long v = atol(e); if (v >= 256 && v <= (1 << 22)) walker_cap = (u64)v; }
    const u64 walkers = s < walker_cap ? s : walker_cap;
    const unsigned wblocks = (unsigned)((walkers + 255) / 256);
    const u64 log_chunks = n / (IDX_CHUNK - 64) + (u64)wblocks * 4 + 2;     // a closed chunk wastes < 64 entries; every wave may leave one open
    BWTS_TRY(arena_reserve(ctx, align_up(n * 4, 256) + radix_tile_hist_bytes(n) + scan_temp_bytes(n) + 24 * s4 +
                                    align_up(node_cap * sizeof(CycleRec), 256) + align_up(node_cap * slot, 256) + align_up(n, 256) +
                                    (mark == MARK_LOG ? align_up(log_chunks * IDX_CHUNK * 4, 256) + align_up(log_chunks * 4, 256) + align_up(n * 4 + (4ull << IDX_RANGE_LOG2), 256) + (1 << 20) : 0) +
                                    (1 << 16)));
    u32 *LF = arena_array<u32>(ctx, n);
    u32 *tile_hist = (u32 *)arena_alloc(ctx, radix_tile_hist_bytes(n));
    void *scan_temp = arena_alloc(ctx, scan_temp_bytes(n));
    u32 *node[12];
    for (int i = 0; i < 12; i++) node[i] = arena_array<u32>(ctx, node_cap);
    LrMin *lrmin[2] = {arena_array<LrMin>(ctx, node_cap), arena_array<LrMin>(ctx, node_cap)};
    LrSum *lrsum[2] = {arena_array<LrSum>(ctx, node_cap), arena_array<LrSum>(ctx, node_cap)};
    CycleRec *d_recs = (CycleRec *)arena_alloc(ctx, node_cap * sizeof(CycleRec));
    u8 *seg = arena_array<u8>(ctx, node_cap * slot);
    const bool bytemark = mark == MARK_BYTEMAP;
    u8 *marks = bytemark ? arena_array<u8>(ctx, n) : nullptr;
    const u32 nbuckets = (u32)((n + (1ull << IDX_RANGE_LOG2) - 1) >> IDX_RANGE_LOG2);
    u32 *idxlog = mark == MARK_LOG ? arena_array<u32>(ctx, log_chunks * IDX_CHUNK) : nullptr;
    u32 *chunk_fill = mark == MARK_LOG ? arena_array<u32>(ctx, log_chunks) : nullptr;
    u32 *bucket_data = mark == MARK_LOG ? arena_array<u32>(ctx, (u64)nbuckets << IDX_RANGE_LOG2) : nullptr;
    u32 *bucket_fill = mark == MARK_LOG ? arena_array<u32>(ctx, (u64)IDX_MAX_BUCKETS * IDX_FILL_STRIDE) : nullptr;
    u32 *bucket_seen = mark == MARK_LOG ? arena_array<u32>(ctx, 2 * IDX_MAX_BUCKETS) : nullptr;   // counts, then deficit flags
    u32 *deficient = bucket_seen ? bucket_seen + IDX_MAX_BUCKETS : nullptr;
    if (!LF || !tile_hist || !scan_temp || !node[11] || !lrmin[1] || !lrsum[1] || !d_recs || !seg || (bytemark && !marks) ||
        (mark == MARK_LOG && (!idxlog || !chunk_fill || !bucket_data || !bucket_fill || !bucket_seen)))
        return BWTS_E_NOMEM;
    if (bytemark) HIPC(hipMemsetAsync(marks, 0, n, ctx->stream));
    u32 *nxt = node[0], *seglen = node[1], *segmin = node[2], *segoff = node[3];
    u32 *d_opos = node[4], *d_wrap = node[5], *d_clen = node[6];
    u32 *dist = node[7], *min_dist = node[8], *end_by_leader = node[9];
    u32 *tmp_idx = node[10], *tmp_val = node[11];

    u64 *hC = ctx->h_small + 1024;          // symbol boundaries C[0..256] (unbwts.c:38-43); filled below
    u64 *dC = ctx->d_small + 1024;

    // stable LF map (unbwts.c:50-52)
    {
        SpanGuard sg(ctx, BWTS_K_LF_BUILD, n, 5 * n);
        lf_hist_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in, n, tile_hist);
        BWTS_TRY(radix_column_scan(ctx, tile_hist, tiles, scan_temp));
        // the scanned table's first row is C itself: no separate histogram sweep, no host round trip before the walk
        ctab_from_tiles_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(tile_hist, n, dC);
        lf_rank_kernel<<<dim3((unsigned)tiles), dim3(LF_THREADS), 0, ctx->stream>>>(d_in, n, tile_hist, LF);
        HIPC(hipGetLastError());
    }

    // the walk: marks, segment symbols, reduced list
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
    {
        SpanGuard sg(ctx, BWTS_K_WALK, n, 6 * n);
        if (mark == MARK_BYTEMAP)
            walk_record_kernel<MARK_BYTEMAP><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, marks, idxlog, s, node_cap, g, slot, dC, seg, nxt, seglen,
                                                                                          segmin, segoff, ticket, ticket + 3, ticket + 4, ticket + 5, chunk_fill, log_chunks, nbuckets, bucket_seen);
        else if (mark == MARK_SENTINEL)
            walk_record_kernel<MARK_SENTINEL><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, marks, idxlog, s, node_cap, g, slot, dC, seg, nxt, seglen,
                                                                                           segmin, segoff, ticket, ticket + 3, ticket + 4, ticket + 5, chunk_fill, log_chunks, nbuckets, bucket_seen);
        else
            walk_record_kernel<MARK_LOG><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(LF, marks, idxlog, s, node_cap, g, slot, dC, seg, nxt, seglen,
                                                                                      segmin, segoff, ticket, ticket + 3, ticket + 4, ticket + 5, chunk_fill, log_chunks, nbuckets, bucket_seen);
        HIPC(hipGetLastError());
    }
    // virtual nodes join the reduced list: its size is only known now
    BWTS_TRY(read_small(ctx, 1024, 257));      // C for the host-side steps below
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

    // elements in splitter-free cycles
    char *ub = nullptr;
    BWTS_TRY(aux_reserve(ctx, 2 * align_up(UNV_CAP * 4, 256) + align_up(UNV_CAP, 256), &ub));
    u32 *uidx = (u32 *)ub, *ulf = (u32 *)(ub + align_up(UNV_CAP * 4, 256));
    {
        SpanGuard sg(ctx, BWTS_K_OTHER, n, 4 * n);
        u64 blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
        if (mark == MARK_LOG) {
            HIPC(hipMemsetAsync(bucket_fill, 0, (size_t)nbuckets * IDX_FILL_STRIDE * sizeof(u32), ctx->stream));
            const int bm_bytes = (int)((1u << IDX_RANGE_LOG2) / 8);
            BWTS_TRY(ensure_dyn_lds(ctx, (const void *)unvisited_from_buckets_kernel, (size_t)bm_bytes));
            BWTS_TRY(ensure_dyn_lds(ctx, (const void *)bucket_indices_kernel, bucket_indices_lds_bytes(IDX_MAX_BUCKETS)));
            bucket_deficit_kernel<<<dim3((nbuckets + 255) / 256), dim3(256), 0, ctx->stream>>>(bucket_seen, nbuckets, n, deficient);
            bucket_indices_kernel<<<dim3((unsigned)log_chunks), dim3(IDX_THREADS), bucket_indices_lds_bytes(nbuckets), ctx->stream>>>(
                idxlog, chunk_fill, nbuckets, deficient, bucket_fill, bucket_data);
            unvisited_from_buckets_kernel<<<dim3(nbuckets), dim3(1024), bm_bytes, ctx->stream>>>(deficient, bucket_fill, bucket_data, LF, n, uidx, ulf, UNV_CAP,
                                                                                                ticket + 1);
        } else if (mark == MARK_BYTEMAP)
            collect_unvisited_kernel<MARK_BYTEMAP><<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(LF, marks, n, uidx, ulf, UNV_CAP, ticket + 1);
        else
            collect_unvisited_kernel<MARK_SENTINEL><<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(LF, marks, n, uidx, ulf, UNV_CAP, ticket + 1);
        HIPC(hipGetLastError());
    }

    // reduced-list ranking on the device
    const int R = [&] { int b = 0; for (u64 x = s_all; x; x >>= 1) b++; return b; }();   // 2^R > s >= any cycle's node count
    int cur = 0, sc = 0;
    {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, s_all, 0);
        const int gb = grid1(s_all);
        lr_init_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, nxt, segmin, lrmin[0]);
        for (int r = 0; r < R; r++, cur ^= 1)
            lr_jump_min_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, lrmin[cur], lrmin[cur ^ 1]);
        lr_cut_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, nxt, seglen, lrmin[cur], lrsum[0]);
        for (int r = 0; r < R; r++, sc ^= 1)
            lr_jump_sum_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, lrsum[sc], lrsum[sc ^ 1]);
        lr_finish_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(s_all, lrmin[cur], lrsum[sc], segmin, segoff, dist, min_dist, d_recs, ticket + 2);
        HIPC(hipGetLastError());
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
                    if (mark == MARK_SENTINEL && n == 0x100000000ull) { *ambiguous = true; return BWTS_OK; }   // see the length check below
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
            if (mark == MARK_SENTINEL && n == 0x100000000ull) { *ambiguous = true; return BWTS_OK; }
            return BWTS_E_INTERNAL;
        }
    }
    HIPC(hipMemcpyAsync(tmp_idx, h_lidx.data(), kc * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipMemcpyAsync(tmp_val, h_lend.data(), kc * 4, hipMemcpyHostToDevice, ctx->stream));
    {
        SpanGuard sg(ctx, BWTS_K_LISTRANK, s_all, 0);
        scatter_u32_kernel<<<dim3(grid1(kc)), dim3(256), 0, ctx->stream>>>(tmp_idx, tmp_val, kc, end_by_leader);
        lr_place_kernel<<<dim3(grid1(s_all)), dim3(256), 0, ctx->stream>>>(s_all, lrmin[cur], lrsum[sc], dist, min_dist, end_by_leader, d_opos, d_wrap, d_clen);
        HIPC(hipGetLastError());
    }

    // the recorded segments go to their places in the text (unbwts.c:73-82)
    {
        SpanGuard sg(ctx, BWTS_K_WALK_EMIT, n, 2 * n);
        int tpn_log2 = g - 4;                           // one 16-symbol chunk per thread at the expected segment length (G)
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
    // how visited entries are recorded: index log (default), or the two mark forms (BWTS_INV_MARK=sentinel|bytemap,
    // BWTS_BYTEMARK=1: tests, and the fallback chain below)
    int mark = MARK_LOG;
    const char *me = getenv("BWTS_INV_MARK");
    if (me && !strcmp(me, "sentinel")) mark = MARK_SENTINEL;
    if ((me && !strcmp(me, "bytemap")) || getenv("BWTS_BYTEMARK")) mark = MARK_BYTEMAP;
    BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, splitter_log2(n), mark, &retry, &ambiguous));
    if (ambiguous) {            // sentinel marks only, n = 2^32: 0xffffffff was a real entry of a splitter-free cycle
        mark = MARK_BYTEMAP;
        BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, splitter_log2(n), mark, &retry, &ambiguous));
    }
    if (retry) {
        if (mark == MARK_LOG) mark = MARK_SENTINEL;        // adversarial LF: keep the retry on the simplest marks
        BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, 0, mark, &retry, &ambiguous));
        if (ambiguous) BWTS_TRY(inverse_attempt(ctx, d_in, n, d_out, 0, MARK_BYTEMAP, &retry, &ambiguous));
        if (retry || ambiguous) return BWTS_E_INTERNAL;
    }
    return BWTS_OK;
}
