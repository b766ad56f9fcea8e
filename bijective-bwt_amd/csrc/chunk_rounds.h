// chunk_rounds.h -- the rounds after round 0 when many elements are tied, third form: CHUNKS.  Included by forward.hip.
//
// dense_rounds() (forward.hip) runs a round as three n-sized launches -- the round kernel leaves (position, head, state) for
// every list element, a count sweep reads the states, the compaction reads all three again -- with two host round trips, and a
// workgroup looks at DG_SPAN list slots to decide DG_OWN of them (a quarter of its loads are halo).  Here the list is cut ONCE
// into chunks at group boundaries, and a chunk belongs to one workgroup in every round:
//   * groups never straddle chunks, so a round needs no halo and no global compaction: the workgroup walks its chunk in tiles of
//     CH_TILE slots cut at group starts, orders every group in LDS exactly like dense_round_kernel, and writes the members that
//     stay tied back IN PLACE, compacted to the front of the chunk (the write cursor never passes the read cursor);
//   * the rank array is only read by the round kernel: the new ranks of the elements whose rank changed go to a per-chunk move
//     list of records, which chunk_apply_records_kernel applies when the round's gathers are all done (a round's keys must come from
//     one version of the ranks);
//   * the grid is one workgroup per chunk in every round and the sizes live on the device, so the host has nothing to read back
//     between rounds except "is anything left / did anything split": once no larger group is left it enqueues two rounds per sync.
// The one-off order by smallest position covers the groups of up to CH_CAP members; the larger ones are put behind them, unordered.
// Of those, groups of up to CH_GROUP_MAX (= a tile) members go to chunks of their own kind at once -- WIDE chunks, which a second
// instantiation of the round kernel handles: every tile ordered by one segmented bitonic sort of the workgroup in LDS instead of
// by counting inside each group; a WIDE chunk whose groups have all fallen to CH_CAP members or fewer is taken over by the
// counting instantiation.  Groups of more than CH_GROUP_MAX members form the BIG LIST, which goes through the sort-based round
// (gather, two radix sorts, regroup -- the larger-group path of dense_rounds on a dense list); whatever falls to CH_GROUP_MAX
// members or fewer leaves it as new chunks appended behind the existing ones.  Groups only ever split, so an element is appended
// at most once and the chunk store never outgrows the list.
#pragma once

#ifndef CH_THREADS
#define CH_THREADS 512
#endif
#ifndef CH_ITEMS
#define CH_ITEMS   4
#endif
#define CH_TILE    (CH_THREADS * CH_ITEMS)
#define CH_CAP     DG_CAP
#define CH_WORDS   (CH_TILE / 64)
#define CH_WORDS_BACK ((CH_CAP + 63) / 64)
#ifndef CH_MIN_WAVES
#define CH_MIN_WAVES 8
#endif
#ifndef CH_FS
#define CH_FS      256
#endif
// (factors whose data a workgroup keeps in LDS: natural data has a dozen or two)
#define CH_GROUP_MAX CH_TILE             // largest group a chunk may hold (a tile of its own in a WIDE chunk)
#define CH_MIN_LIST 65536ull            // shorter lists keep the tile form (dense_rounds)
#define CH_SLOTS   4                    // result slots of rounds in flight
#define CH_SLOT_WORDS 16
#define SM_CHSLOT  (SM_DGCNT + 16)      // CH_SLOTS x CH_SLOT_WORDS words inside the dense rounds' counter block
enum { CHS_SPLIT = 0, CHS_ERR = 1, CHS_TOTAL = 2, CHS_EXIT = 3, CHS_STAY = 4, CHS_BIGGROUPS = 5, CHS_PARKED = 8 };

// ---- PARKED CHAINS (round 4) ---------------------------------------------------------------------------------------------------
// Inside a repeated stretch the copies tie position by position: the tied group A = {x_1 .. x_s} is followed, one position on, by the
// tied group A' = {x_1 + 1 .. x_s + 1}, and so on to the end of the stretch.  All x_i share their first symbol, so the order of A's
// members IS the order of A''s: refining A every round, as the rounds before did (most of a text's tied list sits in such stretches
// for a dozen rounds), repeats what A''s refinement already says.  So: a group whose members' cyclic successors (distance 1) form
// exactly one other tied group of the same size -- all successor ranks equal, that group's size equal to its own, no member at its
// factor's last position -- is not refined.  It PARKS: its members leave the list for good and set a flag in position space.  What a
// round does to a member q of a group that is refined (new rank = old + delta; alone now; new group size) is INDUCED onto the
// positions q - 1, q - 2, ... while they carry the flag: the chain behind q, which by construction holds the corresponding members of
// the groups parked behind q's.  Every chain ends in a group that is refined (the one holding a factor's last position never parks),
// so nothing waits for ever, and a round in which no refined group splits still proves what it proved before: equal infinite words.
// Sound because the partition only ever gets FINER than the classic rounds' (a parked group receives its leader's refinement, which
// looks one symbol further per chain link), and doubling tolerates ranks that are finer than h-consistent.
// pinfo[x] (one byte per text position): low 7 bits = size of x's group while x is tied (127: 127 or more; 0: not tied, or not known:
// members of groups that only the WIDE kernel / the big list have handled so far), bit 7 = parked.  Like rank[], pinfo is only READ by
// the round kernels (except for the parked flag, which no reader of the sizes looks at); the record pass applies the changes.
// A round's record of a member (what chunk_apply_moves_kernel's (new rank, position) pairs were): position, delta, new size, alone.
#define CH_PARK_MAX 126u                 // largest group that may park (its size must be told apart from "127 or more")
#define CH_REC(pos, delta, size7, alone, eqb) ((u64)(pos) | ((u64)(delta) << 32) | ((u64)(size7) << 44) | ((u64)((alone) ? 1u : 0u) << 51) | ((u64)(eqb) << 53))
#define CH_REC_SELF (1ull << 52)         // (the final lay-out of equal words) the member's own byte and suffix-array slot are written by the record pass too
#define CH_SHORT_CHAIN 8                 // chain positions the record's own lane walks before it hands the chain to a wave (0: every chain to a wave -- measured: 58 against 38 ms of record passes per text(2^30) forward)
static_assert(CH_CAP * 4 <= CH_TILE, "a tile must hold several whole groups");
static_assert(16 + CH_SLOTS * CH_SLOT_WORDS <= DG_CNT_BIG + DG_CNT_SPREAD, "result slots live in the dense rounds' counter block");

// WIDE chunks -- those that hold a group of more than CH_CAP members (up to CH_GROUP_MAX: what leaves the big list) -- are handled by a
// second instantiation of the kernel, which orders every tile with one segmented bitonic sort of the whole workgroup over
// (group's first slot, rank at h[, ranks at 2h and 3h], slot) instead of counting inside each group: N slots (a power of two >= the
// tile), the slots behind the tile sort behind it.  pay = group's first slot << 16 | slot.
template <int NKEYS>
__device__ __forceinline__ void chunk_bitonic_sort(u32 *key, u64 *key23, u32 *pay, u32 N, int tid)
{
    for (u32 k = 2; k <= N; k <<= 1)
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 q = (u32)tid; q < N / 2; q += CH_THREADS) {
                const u32 i = ((q & ~(j - 1u)) << 1) | (q & (j - 1u)), l = i | j;
                const bool asc = (i & k) == 0;
                const u32 ka = key[i], kb = key[l], pa = pay[i], pb = pay[l];
                const u32 ga = pa >> 16, gb = pb >> 16;
                bool gt;
                if (NKEYS == 3) {
                    const u64 xa = key23[NKEYS == 3 ? i : 0], xb = key23[NKEYS == 3 ? l : 0];
                    gt = ga != gb ? ga > gb : ka != kb ? ka > kb : xa != xb ? xa > xb : pa > pb;
                    if (gt == asc) { key23[NKEYS == 3 ? i : 0] = xb; key23[NKEYS == 3 ? l : 0] = xa; }
                } else {
                    gt = ga != gb ? ga > gb : ka != kb ? ka > kb : pa > pb;
                }
                if (gt == asc) { key[i] = kb; key[l] = ka; pay[i] = pb; pay[l] = pa; }
            }
            // a stage with j <= 64 stays inside blocks of 128 slots, and pair q belongs to block q / 64: wave w only ever touches blocks
            // w and w + 8.  Between two such stages the wave's own order is enough; the workgroup meets only around the wider ones.
            const u32 jn = j > 1 ? j >> 1 : k;          // the next stage's distance (k: the first stage of the next, doubled k)
            if (j > 64 || jn > 64) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
        }
    __syncthreads();
}

// chunk c = list slots [cstart[c], cstart[c] + ccount[c]); the region up to cstart[c + 1] (or the store's tail) is its own.
// The kernel is bound by its VALU instructions as much as by memory (PMC on the first version: ~350 per element, VALU busy 51 %,
// waves waiting 65 % of their cycles), so everything per element is 32-bit arithmetic: positions are u32 (n <= 2^32; lengths are
// kept modulo 2^32, which the wrap-around of the cyclic successor absorbs); the factor of a position comes from a 256-entry
// directory over the positions' top bits plus (nearly always) one comparison, and its start, length and the steps reduced
// modulo the length from one 16-byte LDS read (FSL: at most CH_FS factors -- natural data has a dozen or two; inputs with more
// take the instantiation with the general 64-bit arithmetic); and a slot's group extent comes from the wave's own __ballot word
// (which is exactly the 64 slots of its lanes) plus two per-word neighbour values, not from a bit search over LDS per lane.
template <bool CYCLIC, int NKEYS, bool FSL /* cyclic, at most CH_FS factors: their data sits in LDS */, bool WIDE /* the chunks flagged in cwide, and only those */,
          bool PARK = false /* parked chains (needs CYCLIC and FSL): its own instantiation, so that the default one pays no register for it */>
__global__ __launch_bounds__(CH_THREADS, WIDE ? 4 : CH_MIN_WAVES) void chunk_round_kernel(u32 *idx, u32 *head, const u32 *__restrict__ cstart, u32 *__restrict__ ccount,
                                                                 u8 *__restrict__ cwide, u64 *__restrict__ mv, u32 *__restrict__ mvcount,
                                                                 const u32 *__restrict__ rank, u64 n, u64 h,
                                                                 const u32 *__restrict__ fstart, u64 k,
                                                                 PrevSym prev, u8 *__restrict__ out, unsigned long long *__restrict__ result,
                                                                 u8 *pinfo /* null: no parked chains */, u32 park_on /* groups may park in this round */)
{
    static_assert(!PARK || (CYCLIC && FSL), "parked chains: cyclic sort with the factors in LDS");
    const bool park = PARK && !WIDE && park_on != 0;
    __shared__ u32 hd[CH_TILE];              // group heads of the tile
    __shared__ u32 key[CH_TILE];             // successor ranks
    __shared__ u64 key23[NKEYS == 3 ? CH_TILE : 1];
    __shared__ u64 startm[CH_WORDS];         // bit = a group starts at this slot
    __shared__ u64 keepm[CH_WORDS];          // bit = the element sorted into this slot stays tied
    __shared__ u32 kpre[CH_WORDS];
    __shared__ uint4 ftab[FSL ? CH_FS + 1 : 1];         // per factor: start, length (mod 2^32), the steps h and 2h modulo the length
    __shared__ u32 fhm3[FSL ? CH_FS : 1];               // ... and 3h
    __shared__ u32 fdir[FSL ? 256 : 1];                 // factor that holds position b << dsh: a lookup starts there
    __shared__ u32 hd2[WIDE ? CH_TILE : 1];  // (WIDE) the heads, while hd carries the sort's payload
    __shared__ u32 s_surv, s_nmv, s_split, s_err, s_wide, s_npark;
    const int tid0 = threadIdx.x;
    const u32 c = blockIdx.x;
    if ((cwide[c] != 0) != WIDE) return;
    const u64 base = (u64)(u32)__builtin_amdgcn_readfirstlane((int)cstart[c]);
    const u32 cnt = (u32)__builtin_amdgcn_readfirstlane((int)ccount[c]);
    if (cnt == 0) { if (tid0 == 0) mvcount[c] = 0; return; }
    if (tid0 == 0) { s_nmv = 0; s_split = 0; s_err = 0; s_wide = 0; s_npark = 0; }
    // the step, wave-uniform.  Cyclic with the factors in LDS: per factor the step(s) reduced modulo its length (a division only
    // for factors shorter than the step: the short ones at the text's end), and a 256-entry directory over the positions' top bits
    // so that finding a position's factor is one table read and (nearly always) one comparison.  Suffixes: p + j h < n <=> p < nhj.
    u32 nh1 = 0, nh2 = 0, nh3 = 0;
    const u32 h32 = (u32)h, k32 = (u32)k;
    int dsh = 0;
    if (FSL) {
        for (u32 f = tid0; f < k32; f += CH_THREADS) {
            const u64 s0 = fstart[f], L = (f + 1 < k32 ? (u64)fstart[f + 1] : n) - s0;
            ftab[f] = make_uint4((u32)s0, (u32)L, (u32)(h < L ? h : h % L), NKEYS == 3 ? (u32)(2 * h < L ? 2 * h : (2 * h) % L) : 0u);
            fhm3[f] = NKEYS == 3 ? (u32)(3 * h < L ? 3 * h : (3 * h) % L) : 0u;
        }
        { int bl = 0; for (u64 x = n - 1; x; x >>= 1) bl++; dsh = bl > 8 ? bl - 8 : 0; }
        if (tid0 < 256) {
            const u64 want = (u64)tid0 << dsh;
            u32 lo = 0, hi = k32 - 1;
            while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if ((u64)fstart[mid] <= want) lo = mid; else hi = mid - 1; }
            fdir[tid0] = lo;
        }
    } else if (!CYCLIC) {
        nh1 = h < n ? (u32)(n - h) : 0u;            // (h >= 1, so n - h fits)
        if (NKEYS == 3) { nh2 = 2 * h < n ? (u32)(n - 2 * h) : 0u; nh3 = 3 * h < n ? (u32)(n - 3 * h) : 0u; }
    }
    u32 rp = 0, wp = 0;                      // read / write cursors inside the chunk (uniform)
#ifdef CH_PROFILE
    long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
#define CH_MARK(i) do { const long long tn__ = clock64(); pt[i] += tn__ - tprev; tprev = tn__; } while (0)
#else
#define CH_MARK(i) do { } while (0)
#endif
    while (rp < cnt) {
        // (an opaque copy of the thread id per iteration: left to itself the compiler hoists every address it can form from
        // tid out of the loop and keeps ~80 registers of them alive through the whole body -- 132 VGPRs instead of 48)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const u32 wv = (u32)__builtin_amdgcn_readfirstlane(tid >> 6);
        const u32 len = cnt - rp < CH_TILE ? cnt - rp : (u32)CH_TILE;
        const bool final_tile = rp + len == cnt;
        u32 myh[CH_ITEMS], myp[CH_ITEMS];
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++) {
            const u32 sl = (u32)j * CH_THREADS + tid;
            myh[j] = sl < len ? head[base + rp + sl] : 0u;
            myp[j] = sl < len ? idx[base + rp + sl] : 0u;
        }
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++) hd[j * CH_THREADS + tid] = myh[j];
        if (tid < CH_WORDS) keepm[tid] = 0;
        __syncthreads();
        CH_MARK(0);
        // successor ranks: they depend on the positions alone, so the gathers are issued now and fly while the group extents are
        // worked out (slots of a group that is left for the next tile are gathered for nothing)
        // (Round 4 tried all twelve gathers of a lane in one straight run of loads -- the compiler drains them item by item, three in
        // flight, because the next item's factor search stands between: 8 % SLOWER, five more spilled registers and no shorter waits.)
        u32 my_key[CH_ITEMS];
        u64 my_key23[NKEYS == 3 ? CH_ITEMS : 1];
        u32 ppos[CH_ITEMS];                      // position of the previous symbol, T[cprev(p)] (mk_bwts_sa.c:172-188), while the factor is at hand
        u32 s1r[PARK ? CH_ITEMS : 1], s1i[PARK ? CH_ITEMS : 1];        // parking: rank and pinfo byte of the cyclic successor at distance 1 (s1i = 0x100: p ends its factor)
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++) {
            my_key[j] = 0; ppos[j] = 0;
            if (PARK) { s1r[PARK ? j : 0] = 0; s1i[PARK ? j : 0] = 0x100u; }
            if (NKEYS == 3) my_key23[NKEYS == 3 ? j : 0] = 0;
            if ((u32)j * CH_THREADS + tid >= len) continue;
            const u32 p = myp[j];
            if (CYCLIC) {
                u32 q1, q2 = 0, q3 = 0;
                if (FSL) {
                    // offset in the factor + step, minus the factor's length when the sum passes it (a carry out of 32 bits passes
                    // it too; a length of 2^32 is kept as 0: the subtraction then does nothing and the wrapped sum is already right)
                    u32 f = fdir[p >> dsh];
                    while (f + 1 < k32 && ftab[f + 1].x <= p) f++;
                    const uint4 ft = ftab[f];
                    const u32 s0 = ft.x, L = ft.y, dd = p - s0;
                    ppos[j] = dd ? p - 1u : s0 + L - 1u;
                    u32 o = dd + ft.z;
                    o = (o < dd || o >= L) ? o - L : o;
                    q1 = s0 + o;
                    if (PARK && park && dd + 1u != L) { s1r[PARK ? j : 0] = rank[p + 1u]; s1i[PARK ? j : 0] = (u32)pinfo[p + 1u]; }
                    if (NKEYS == 3) {
                        u32 o2 = dd + ft.w, o3 = dd + fhm3[f];
                        o2 = (o2 < dd || o2 >= L) ? o2 - L : o2;
                        o3 = (o3 < dd || o3 >= L) ? o3 - L : o3;
                        q2 = s0 + o2; q3 = s0 + o3;
                    }
                } else {
                    const u64 f = factor_of(fstart, k, (u64)p);
                    const u64 s0 = fstart[f], e1 = factor_end(fstart, k, n, f);
                    ppos[j] = p == s0 ? (u32)(e1 - 1) : p - 1u;
                    q1 = (u32)cyclic_successor(p, s0, e1 - s0, h);
                    if (NKEYS == 3) { q2 = (u32)cyclic_successor(p, s0, e1 - s0, 2 * h); q3 = (u32)cyclic_successor(p, s0, e1 - s0, 3 * h); }
                }
                my_key[j] = rank[q1];
                if (NKEYS == 3) { const u32 r2 = rank[q2], r3 = rank[q3]; my_key23[NKEYS == 3 ? j : 0] = ((u64)r2 << 32) | r3; }
            } else {
                my_key[j] = p < nh1 ? rank[p + h32] + 1u : 0u;
                if (NKEYS == 3) {
                    const u32 r2 = p < nh2 ? rank[p + 2u * h32] + 1u : 0u, r3 = p < nh3 ? rank[p + 3u * h32] + 1u : 0u;
                    my_key23[NKEYS == 3 ? j : 0] = ((u64)r2 << 32) | r3;
                }
            }
        }
        // group starts: the 64 slots of a wave's item j are exactly word j * 8 + wave of the tile
        u64 stm[CH_ITEMS];
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++) {
            const u32 sl = (u32)j * CH_THREADS + tid;
            // slot 0 starts a group by construction (chunks and tiles are cut at group starts); slots past the end count as starts
            const bool st = sl >= len || sl == 0 || myh[j] != hd[sl - 1];
            stm[j] = __ballot(st);
            if (lane == 0) startm[(u32)j * (CH_THREADS / 64) + wv] = stm[j];
        }
        __syncthreads();
        const u64 le = lane == 63 ? ~0ull : (2ull << lane) - 1ull;       // slots of the word at or below mine
        u32 plen = len;
        u32 dst[CH_ITEMS], newhead[CH_ITEMS], dl[CH_ITEMS], nsz[CH_ITEMS], eqb[CH_ITEMS];      // dl: new rank - old rank; nsz: size of the new group (0 alone, 127 = 127 or more); eqb: place in it
        bool alone[CH_ITEMS], moved[CH_ITEMS], act[CH_ITEMS], prk[CH_ITEMS];
        u32 split_here = 0;
        if constexpr (WIDE) {
            // ---- every group of the tile ordered by one segmented sort of the workgroup ----
            if (!final_tile) {
                int w = CH_WORDS - 1;
                u64 m = startm[w];
                while (m == 0 && w > 0) { w--; m = startm[w]; }
                plen = (u32)w * 64u + 63u - (u32)__clzll((long long)m);
            }
            plen = (u32)__builtin_amdgcn_readfirstlane((int)plen);
            if (plen == 0) {
                // one group fills the tile: it ends exactly here (the next slot shows another head), or it is too large to be in a chunk
                if (head[base + rp + len] != hd[0]) plen = len;
                else { if (tid == 0) s_err = 1; break; }
            }
            u32 N = 64; while (N < plen) N <<= 1;
            // every slot's group: the last start at or below it
            u32 gsl[CH_ITEMS];
#pragma unroll
            for (int j = 0; j < CH_ITEMS; j++) {
                const u32 sl = (u32)j * CH_THREADS + tid;
                gsl[j] = 0;
                if (sl < plen) {
                    u32 w = (u32)j * (CH_THREADS / 64) + wv;
                    u64 m = stm[j] & le;
                    while (m == 0ull) { w--; m = startm[w]; }          // (slot 0 starts a group: the walk ends)
                    gsl[j] = w * 64u + 63u - (u32)__clzll((long long)m);
                }
            }
            __syncthreads();                            // (every read of the heads in hd and of the start words is done)
#pragma unroll
            for (int j = 0; j < CH_ITEMS; j++) {
                const u32 sl = (u32)j * CH_THREADS + tid;
                hd2[WIDE ? sl : 0] = myh[j];
                if (sl < N) {
                    const bool real = sl < plen;
                    key[sl] = real ? my_key[j] : 0xffffffffu;
                    if (NKEYS == 3) key23[NKEYS == 3 ? sl : 0] = real ? my_key23[NKEYS == 3 ? j : 0] : ~0ull;
                    hd[sl] = ((real ? gsl[j] : 0xffffu) << 16) | sl;
                }
            }
            __syncthreads();
            chunk_bitonic_sort<NKEYS>(key, key23, hd, N, tid);
            // sorted slot t of this thread: whose element, which group, does a run of equal keys start / end here?
            bool rstart[CH_ITEMS], rend[CH_ITEMS];
            u32 src[CH_ITEMS], gst[CH_ITEMS];
            u64 rsm[CH_ITEMS];
#pragma unroll
            for (int j = 0; j < CH_ITEMS; j++) {
                const u32 t = (u32)j * CH_THREADS + tid;
                rstart[j] = true; rend[j] = true; src[j] = 0; gst[j] = 0;
                if (t < plen) {
                    const u32 pt = hd[t], kt = key[t];
                    const u64 xt = NKEYS == 3 ? key23[NKEYS == 3 ? t : 0] : 0ull;
                    gst[j] = pt >> 16; src[j] = pt & 0xffffu;
                    if (t > gst[j]) rstart[j] = key[t - 1] != kt || (NKEYS == 3 && key23[NKEYS == 3 ? t - 1 : 0] != xt);
                    if (t + 1 < plen && (hd[t + 1] >> 16) == gst[j]) rend[j] = key[t + 1] != kt || (NKEYS == 3 && key23[NKEYS == 3 ? t + 1 : 0] != xt);
                }
                rsm[j] = __ballot(rstart[j]);
            }
            __syncthreads();                            // (the start words of the groups have been read by everyone: they now hold the runs')
#pragma unroll
            for (int j = 0; j < CH_ITEMS; j++)
                if (lane == 0) startm[(u32)j * (CH_THREADS / 64) + wv] = rsm[j];
            // the elements' positions through LDS (the keys are not needed any more)
#pragma unroll
            for (int j = 0; j < CH_ITEMS; j++) {
                const u32 sl = (u32)j * CH_THREADS + tid;
                if (sl < plen) key[sl] = myp[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < CH_ITEMS; j++) {
                const u32 t = (u32)j * CH_THREADS + tid;
                act[j] = t < plen;
                dst[j] = t; newhead[j] = 0; alone[j] = false; moved[j] = false; ppos[j] = 0; prk[j] = false; dl[j] = 0; nsz[j] = 0; eqb[j] = 0;
                if (!act[j]) continue;
                u32 w = (u32)j * (CH_THREADS / 64) + wv;
                u64 m = rsm[j] & le;
                while (m == 0ull) { w--; m = startm[w]; }              // (a group's first slot starts a run: the walk ends there at the latest)
                const u32 rbeg = w * 64u + 63u - (u32)__clzll((long long)m);
                const u32 less = rbeg - gst[j];
                const u32 p = key[src[j]];
                myp[j] = p;
                newhead[j] = hd2[WIDE ? gst[j] : 0] + less;
                dl[j] = less;
                eqb[j] = t - rbeg;
                alone[j] = rstart[j] && rend[j];
                // where the next run starts (the start words hold the runs' by now): the run's length, and whether the run is its whole group
                u32 nx;
                {
                    u32 w2 = (u32)j * (CH_THREADS / 64) + wv;
                    u64 ab = rsm[j] & ~le;
                    while (ab == 0ull && w2 + 1 < CH_WORDS) { w2++; ab = startm[w2]; }
                    nx = ab ? w2 * 64u + (u32)__ffsll((unsigned long long)ab) - 1u : plen;
                    if (nx > plen) nx = plen;
                }
                const u32 rlen = nx - rbeg;
                nsz[j] = alone[j] ? 0u : (rlen < 127u ? rlen : 127u);
                const bool whole = less == 0 && (nx >= plen || (hd[nx] >> 16) != gst[j]);
                moved[j] = !whole;                                       // (its rank or its group's size changes: a record either way)
                split_here |= less != 0 ? 1u : 0u;
                if (!alone[j] && t - rbeg >= CH_CAP) s_wide = 1;        // a run of more than CH_CAP members stays: the chunk stays WIDE
                if (CYCLIC && out && (alone[j] || (PARK && moved[j])) && !prev.P) {
                    if (FSL) {
                        u32 f = fdir[p >> dsh];
                        while (f + 1 < k32 && ftab[f + 1].x <= p) f++;
                        const uint4 ft = ftab[f];
                        ppos[j] = p != ft.x ? p - 1u : ft.x + ft.y - 1u;
                    } else {
                        const u64 f = factor_of(fstart, k, (u64)p);
                        const u64 s0 = fstart[f], e1 = factor_end(fstart, k, n, f);
                        ppos[j] = p == s0 ? (u32)(e1 - 1) : p - 1u;
                    }
                }
            }
            CH_MARK(1); CH_MARK(2);
        } else {
        // the group the tile's last start opens may go on in the next tile: it is left for that one
        if (!final_tile) {
            int w = CH_WORDS - 1;
            u64 m = startm[w];
            while (m == 0 && w > 0) { w--; m = startm[w]; }
            plen = (u32)w * 64u + 63u - (u32)__clzll((long long)m);
        }
        plen = (u32)__builtin_amdgcn_readfirstlane((int)plen);        // uniform: keep it (and the cursors) in scalar registers
        if (plen == 0) { if (tid == 0) s_err = 1; break; }          // a group of a whole tile: larger than CH_CAP, cannot be here
        // per word (lane l < CH_WORDS of every wave <-> word l): the last start before it, the first start behind it
        int cin_l = -1, cout_l = -1;
        if (lane < CH_WORDS) {
#pragma unroll
            for (int d = 1; d <= CH_WORDS_BACK; d++)
                if (cin_l < 0 && lane >= d) { const u64 pm = startm[lane - d]; if (pm) cin_l = (lane - d) * 64 + 63 - __clzll((long long)pm); }
#pragma unroll
            for (int d = 1; d <= CH_WORDS_BACK; d++)
                if (cout_l < 0 && lane + d < CH_WORDS) { const u64 nm = startm[lane + d]; if (nm) cout_l = (lane + d) * 64 + __ffsll((unsigned long long)nm) - 1; }
        }
        u32 gs[CH_ITEMS], sz[CH_ITEMS];
        bool bad = false;
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++) {
            const u32 sl = (u32)j * CH_THREADS + tid;
            const u32 w = (u32)j * (CH_THREADS / 64) + wv;
            act[j] = sl < plen;
            gs[j] = 0; sz[j] = 0;
            const int cin = __builtin_amdgcn_readlane(cin_l, (int)w), cout = __builtin_amdgcn_readlane(cout_l, (int)w);
            const u64 below = stm[j] & le, above = stm[j] & ~le;
            const int g = below ? (int)(w * 64u) + 63 - __clzll((long long)below) : cin;
            int e = above ? (int)(w * 64u) + __ffsll((unsigned long long)above) - 1 : cout;
            if (e < 0 || (u32)e > plen) e = (int)plen;
            if (!act[j]) continue;
            if (g < 0 || e - g > CH_CAP) {
                bad = true; act[j] = false;
                if (atomicCAS(&result[5], 0ull, 1ull + c) == 0ull) {       // first failure: where (read by the host under BWTS_ROUND_TRACE)
                    result[6] = ((u64)(u32)g << 32) | (u32)e;
                    result[7] = ((u64)plen << 48) | ((u64)len << 32) | ((u64)rp << 16) | sl;
                }
                continue;
            }
            gs[j] = (u32)g; sz[j] = (u32)(e - g);
        }
        if (bad) s_err = 1;
        CH_MARK(1);
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++)
            if (act[j]) {
                key[j * CH_THREADS + tid] = my_key[j];
                if (NKEYS == 3) key23[NKEYS == 3 ? j * CH_THREADS + tid : 0] = my_key23[NKEYS == 3 ? j : 0];
                // (the heads in hd have been read by everyone: the start words say the rest) link code of this member: its successor's
                // rank when that successor is tied in a group of this group's size, else a value no rank takes among tied elements
                if (PARK && park) hd[j * CH_THREADS + tid] = (sz[j] <= CH_PARK_MAX && (s1i[PARK ? j : 0] & 0x17fu) == sz[j]) ? s1r[PARK ? j : 0] : 0xffffffffu;
            }
        __syncthreads();
        CH_MARK(2);
        // order inside the group by counting (as dense_round_kernel)
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++) {
            const u32 sl = (u32)j * CH_THREADS + tid;
            dst[j] = sl; newhead[j] = 0; alone[j] = false; moved[j] = false; prk[j] = false; dl[j] = 0; nsz[j] = 0; eqb[j] = 0;
            if (!act[j]) continue;
            const u32 g0 = gs[j], gsz = sz[j], mine = key[sl];
            const u32 mylink = (PARK && park) ? hd[sl] : 0xffffffffu;
            u32 unlike = 0;                         // members whose link code differs from mine
            u32 less = 0, eq = 0, eq_before = 0;
            if (NKEYS == 3) {
                const u64 mine23 = key23[NKEYS == 3 ? sl : 0];
                u32 m = 0;
                for (; m + 4 <= gsz; m += 4) {
                    u32 ko[4]; u64 ko23[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) { ko[q] = key[g0 + m + q]; ko23[q] = key23[NKEYS == 3 ? g0 + m + q : 0]; }
                    if (PARK && park) {
#pragma unroll
                        for (int q = 0; q < 4; q++) unlike |= hd[g0 + m + q] ^ mylink;
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const bool same = ko[q] == mine && ko23[q] == mine23;
                        less += (ko[q] < mine || (ko[q] == mine && ko23[q] < mine23)) ? 1u : 0u;
                        eq += same ? 1u : 0u;
                        eq_before += (same && g0 + m + q < sl) ? 1u : 0u;
                    }
                }
                for (; m < gsz; m++) {
                    const u32 ko = key[g0 + m];
                    const u64 ko23 = key23[NKEYS == 3 ? g0 + m : 0];
                    if (PARK && park) unlike |= hd[g0 + m] ^ mylink;
                    const bool same = ko == mine && ko23 == mine23;
                    less += (ko < mine || (ko == mine && ko23 < mine23)) ? 1u : 0u;
                    eq += same ? 1u : 0u;
                    eq_before += (same && g0 + m < sl) ? 1u : 0u;
                }
            } else {
                for (u32 m = 0; m < gsz; m++) {
                    const u32 ko = key[g0 + m];
                    if (PARK && park) unlike |= hd[g0 + m] ^ mylink;
                    less += ko < mine ? 1u : 0u;
                    eq += ko == mine ? 1u : 0u;
                    eq_before += (ko == mine && g0 + m < sl) ? 1u : 0u;
                }
            }
            if (PARK && park && mylink != 0xffffffffu && unlike == 0) {
                // every member's successor lies in ONE tied group of this group's size: the group parks behind it (see PARKED CHAINS)
                prk[j] = true;
                nsz[j] = gsz;
                continue;
            }
            dst[j] = g0 + less + eq_before;
            newhead[j] = myh[j] + less;
            dl[j] = less;
            eqb[j] = eq_before;
            nsz[j] = eq == 1 ? 0u : (eq < 127u ? eq : 127u);
            alone[j] = eq == 1;
            moved[j] = less != 0 || eq < gsz;                       // (its rank or its group's size changes: a record either way)
            split_here |= eq < gsz ? 1u : 0u;
        }
        }
        CH_MARK(3);
        u32 pv[CH_ITEMS];
        bool wr[CH_ITEMS];
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++) {
            // (a member whose cyclic predecessor is parked behind it needs no byte: its whole group has ONE previous byte, in place since
            // round 0 -- ChainCtx::at.  ppos is that predecessor when the factors sit in LDS and no previous-symbol array exists)
            // With parked chains every member of a group that splits writes, alone or not, to its place in its new group's slots
            // (ChainCtx::at says why); without them a member writes once, when it is alone.
            wr[j] = out && act[j] && !prk[j] && (alone[j] || (PARK && moved[j]));
            if (PARK) wr[j] = wr[j] && !(!prev.P && (pinfo[ppos[j]] & 0x80u));
            pv[j] = wr[j] ? (prev.P ? (u32)prev.P[myp[j]] : (u32)prev.T[ppos[j]]) : 0u;
        }
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++)
            if (act[j] && !alone[j] && !prk[j]) atomicOr((unsigned long long *)&keepm[dst[j] >> 6], 1ull << (dst[j] & 63u));
        // members of groups that park: the flag in position space (the group's size stays as it is), and they are gone from the list
        if (PARK && park) {
            u32 np = 0;
#pragma unroll
            for (int j = 0; j < CH_ITEMS; j++)
                if (act[j] && prk[j]) { pinfo[myp[j]] = (u8)(nsz[j] | 0x80u); np++; }
            const u64 any = __ballot(np != 0);
            if (any) {
                u32 tot = np;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) tot += (u32)__shfl_xor((int)tot, d, 64);
                if (lane == 0) atomicAdd(&s_npark, tot);
            }
        }
        // what happened to the members whose rank or group changed: to the chunk's record list (any order)
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++) {
            const bool mvd = act[j] && !prk[j] && (PARK ? (moved[j] || alone[j]) : dl[j] != 0);
            const u64 mm = __ballot(mvd);
            if (mm) {
                const int leader = __ffsll((unsigned long long)mm) - 1;
                u32 b0 = 0;
                if (lane == leader) b0 = atomicAdd(&s_nmv, (u32)__popcll(mm));
                b0 = shfl_t(b0, leader);
                if (mvd) mv[base + b0 + (u32)__popcll(mm & (le >> 1))] = PARK ? CH_REC(myp[j], dl[j], nsz[j], alone[j], eqb[j]) : (((u64)newhead[j] << 32) | (u64)myp[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++)
            if (wr[j]) out[newhead[j] + (PARK ? eqb[j] : 0u)] = (u8)pv[j];
        if (split_here) s_split = 1;
        __syncthreads();
        CH_MARK(4);
        if (tid < 64) {
            const u32 cpop = tid < CH_WORDS ? (u32)__popcll(keepm[tid < CH_WORDS ? tid : 0]) : 0u;
            const u32 inc = wave_scan_inclusive(cpop, OpAdd());
            if (tid < CH_WORDS) kpre[tid] = inc - cpop;
            if (tid == CH_WORDS - 1) s_surv = inc;
        }
        __syncthreads();
        // the members that stay, in sorted order, to the front of the chunk: wp + survivors <= rp + plen, and every load of this
        // tile is behind the barriers above, so nothing unread is overwritten
#pragma unroll
        for (int j = 0; j < CH_ITEMS; j++)
            if (act[j] && !alone[j] && !prk[j]) {
                const u32 w = dst[j] >> 6, b = dst[j] & 63u;
                const u64 lowbits = keepm[w] & ((1ull << b) - 1ull);
                const u32 o = wp + kpre[w] + (u32)__popcll(lowbits);
                idx[base + o] = myp[j];
                head[base + o] = newhead[j];
            }
        wp += (u32)__builtin_amdgcn_readfirstlane((int)s_surv);
        rp += plen;
        __syncthreads();
        CH_MARK(5);
    }
    __syncthreads();
    if (tid0 == 0) {
        ccount[c] = wp;
        mvcount[c] = s_nmv;
        if (PARK && s_npark) atomicAdd(&result[CHS_PARKED], (unsigned long long)s_npark);
        if (WIDE && !s_wide) cwide[c] = 0;          // only groups of up to CH_CAP members are left: the other instantiation takes over
        if (s_split && __hip_atomic_load(&result[CHS_SPLIT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            __hip_atomic_store(&result[CHS_SPLIT], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_err) __hip_atomic_store(&result[CHS_ERR], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef CH_PROFILE
        for (int i = 0; i < 6; i++) atomicAdd(&result[64 + 8 * (c & 63) + i], (unsigned long long)pt[i]);      // spread: one address would serialise
#endif
    }
#undef CH_MARK
}

// ---- the record pass: runs when every gather of the round is done ---------------------------------------------------------------
// A record says what the round did to one member q: rank[q] += delta, pinfo[q] = new size (0: alone).  The same then happens to the
// positions q - 1, q - 2, ... while they carry the parked flag -- the chain behind q (PARKED CHAINS above) -- and a position whose
// member came out alone gets its output byte there (and, in the final lay-out, its suffix-array slot).  A chain never leaves q's
// factor (a parked position is never its factor's last one), so the only position whose previous symbol is not the byte before it
// is the factor's first.  The record's own lane walks up to CH_SHORT_CHAIN positions; longer chains are finished by the workgroup's
// waves, 64 positions per step (pass 2 of the kernel).
struct ChainCtx {
    u32 *rank; u8 *pinfo; PrevSym prev; u8 *out; u32 *SA;
    // (the byte of a chain position that starts its factor is not the byte before it: those few are put right by
    // chunk_factor_heads_kernel when the rounds are over, instead of a factor search per chain here)
    __device__ __forceinline__ u8 byte_before(u32 x) const { return prev.P ? prev.P[x] : prev.T[x ? x - 1u : 0u]; }
    // One chain position.  OUTPUT BYTES.  Kept true for every tied group, refined or parked: its slots [head, head + size) hold its
    // members' previous bytes, in SOME order (round 0's carried bytes start it; every member of a group that splits writes its byte to
    // its place in its new group's slots, alone or not).  A position x whose predecessor x - 1 is parked too then never needs a byte
    // written: the group parked at x - 1 is, member for member, the set of predecessors of x's group, so all of that group's members
    // have the SAME previous byte and any split of its slots holds the right bytes already.  Only the chain's lowest position
    // (`last`) may sit in a group with different bytes: it writes like a member of a refined group.  A member that came out alone leaves no
    // group size to keep (its rank is unique now: no group's successors can all point at it), so pinfo is only written for members
    // that stay tied; the parked flag of a finished chain stays set -- nothing ever walks there again.
    __device__ __forceinline__ void at(u32 x, u32 r, u32 delta, u32 size7, bool alone, bool last, u32 eqb) const
    {
        const u32 nr = r + delta;
        if (delta) rank[x] = nr;
        if (!alone) pinfo[x] = (u8)(size7 | 0x80u);
        if (out && last) out[nr + eqb] = byte_before(x);
        if (alone && SA) SA[nr] = x;
    }
};
// when the rounds are over (ranks final): the output byte of every factor's first position is its factor's last byte (mk_bwts_sa.c:172-188)
__global__ __launch_bounds__(256) void chunk_factor_heads_kernel(const u8 *__restrict__ T, u64 n, const u32 *__restrict__ fstart, u64 k, const u32 *__restrict__ rank,
                                                                 u8 *__restrict__ out)
{
    const u64 f = (u64)blockIdx.x * 256 + threadIdx.x;
    if (f >= k) return;
    const u64 s0 = fstart[f], e1 = f + 1 < k ? (u64)fstart[f + 1] : n;
    out[rank[s0]] = T[e1 - 1];
}
#define CH_REC_POS(e)   ((u32)(e))
#define CH_REC_DELTA(e) ((u32)((e) >> 32) & 0xfffu)
#define CH_REC_SIZE(e)  ((u32)((e) >> 44) & 0x7fu)
#define CH_REC_ALONE(e) ((((e) >> 51) & 1ull) != 0)
#define CH_REC_EQB(e)   ((u32)((e) >> 53))           /* the member's place inside its new group (0 when alone) */

// how many of the positions q - 1, q - 2, ... q - 8 (in this order) carry the parked flag before the first that does not: 0 .. 8
__device__ __forceinline__ u32 chain_peek8(const u8 *__restrict__ pinfo, u32 q)
{
    if (q < 8u) { u32 c = 0; while (c < q && (pinfo[q - 1u - c] & 0x80u)) c++; return c; }
    const u64 a = (u64)(q - 8u) & ~7ull;                       // (pinfo is allocated in multiples of 256 bytes: a + 16 <= its end)
    const u64 lo = *(const u64 *)(pinfo + a), hi = *(const u64 *)(pinfo + a + 8);
    const u32 sh = (u32)((u64)(q - 8u) - a) * 8u;
    const u64 w = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;   // byte 7 = position q - 1 ... byte 0 = position q - 8
    const u64 clear = ~(w | 0x7f7f7f7f7f7f7f7full);            // top bit of a byte set <=> that position is NOT parked
    return clear ? (u32)__clzll((long long)clear) >> 3 : 8u;
}

// pass 1, a lane per record: the member itself and the first CH_SHORT_CHAIN (= 8: one peek) positions of its chain.  Records whose chain goes
// on are kept, compacted to the front of the chunk's own record list (never ahead of the records still to be read); lcount[c] says how many.
// MODE 0: the rounds without parked chains: a record is (new rank << 32 | position) of a member whose rank changed, nothing else to do.
// MODE 1: their final lay-out of equal words: the same records for every member left, with its byte and suffix-array slot.  MODE 2: the above.
template <int MODE>
__global__ __launch_bounds__(256) void chunk_apply_records_kernel(u64 *__restrict__ mv, const u32 *__restrict__ cstart, const u32 *__restrict__ mvcount, ChainCtx cx,
                                                                  u32 *__restrict__ lcount)
{
    if (MODE != 2) {
        const u32 c = blockIdx.x;
        const u32 m = mvcount[c];
        const u64 base = cstart[c];
        // (four records per lane and step, read before any of them is applied: the rank array may alias the records as far as the
        // compiler knows, so one by one every record's load waited for the store before it)
        for (u32 i0 = 0; i0 < m; i0 += 1024) {
            u64 e[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { const u32 i = i0 + (u32)t * 256 + threadIdx.x; e[t] = mv[base + (i < m ? i : m - 1)]; }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (i0 + (u32)t * 256 + threadIdx.x >= m) continue;
                const u32 q = (u32)e[t], nr = (u32)(e[t] >> 32);
                cx.rank[q] = nr;
                if (MODE == 1) {
                    if (cx.out) cx.out[nr] = cx.prev((u64)q);
                    if (cx.SA) cx.SA[nr] = q;
                }
            }
        }
        return;
    }
    static_assert(CH_SHORT_CHAIN == 8, "chain_peek8 looks at eight positions");
    __shared__ u32 s_long;
    const u32 c = blockIdx.x;
    const u32 m = mvcount[c];
    const u64 base = cstart[c];
    const int lane = lane_id();
    if (threadIdx.x == 0) s_long = 0;
    __syncthreads();
    for (u32 i0 = 0; i0 < m; i0 += 256) {
        const u32 i = i0 + threadIdx.x;
        bool lng = false;
        u64 e = 0;
        if (i < m) {
            e = mv[base + i];
            const u32 q = CH_REC_POS(e), delta = CH_REC_DELTA(e), size7 = CH_REC_SIZE(e);
            const bool alone = CH_REC_ALONE(e);
            const u32 cl = cx.pinfo ? chain_peek8(cx.pinfo, q) : 0u;
            u32 nr = 0;
            if (delta || (e & CH_REC_SELF)) { nr = cx.rank[q] + delta; if (delta) cx.rank[q] = nr; }
            if (cx.pinfo && !alone) cx.pinfo[q] = (u8)size7;
            if (e & CH_REC_SELF) {
                if (cx.out) cx.out[nr] = cx.prev((u64)q);
                if (cx.SA) cx.SA[nr] = q;
            }
            if (cl) {
                lng = cl == 8u && q > 8u && (cx.pinfo[q - 9u] & 0x80u);
                // a member that came out alone without changing its rank changes nothing along its chain but the lowest position's byte
                const bool walk = delta != 0 || !alone || cx.SA != nullptr;
                if (walk) {
                    u32 r[8];
#pragma unroll
                    for (u32 t = 0; t < 8; t++) r[t] = t < cl ? cx.rank[q - 1u - t] : 0u;
#pragma unroll
                    for (u32 t = 0; t < 8; t++) if (t < cl) cx.at(q - 1u - t, r[t], delta, size7, alone, !lng && t + 1 == cl, CH_REC_EQB(e));
                } else if (!lng && cx.out) {
                    const u32 x = q - cl;
                    cx.out[cx.rank[x]] = cx.byte_before(x);
                }
            }
        }
        __syncthreads();                              // every record of this step has been read
        const u64 lm = __ballot(lng);
        if (lm) {
            const int leader = __ffsll((unsigned long long)lm) - 1;
            u32 b0 = 0;
            if (lane == leader) b0 = atomicAdd(&s_long, (u32)__popcll(lm));
            b0 = shfl_t(b0, leader);
            if (lng) mv[base + b0 + (u32)__popcll(lm & lanemask_lt())] = e;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) lcount[c] = s_long;
}

// pass 2, a wave per kept record; every wave takes a contiguous share of all chunks' kept records (loff: exclusive sums of lcount, the
// total behind them; one search for the share's first chunk, then a walk): the chain from position q - 1 - CH_SHORT_CHAIN downwards,
// 64 positions per step; flag and rank of a step's positions are fetched together, so a step costs one memory round trip.
__global__ __launch_bounds__(256) void chunk_long_chains_kernel(const u64 *__restrict__ mv, const u32 *__restrict__ cstart, const u32 *__restrict__ loff, u32 nchunks,
                                                                ChainCtx cx)
{
    const int lane = lane_id();
    const u32 total = loff[nchunks];
    const u32 wave0 = (u32)(((u64)blockIdx.x * 256 + threadIdx.x) >> 6), waves = (u32)(((u64)gridDim.x * 256) >> 6);
    const u32 per = (total + waves - 1) / waves;
    const u32 t0 = wave0 * per < total ? wave0 * per : total, t1 = t0 + per < total ? t0 + per : total;
    if (t0 >= t1) return;
    u32 c = 0;
    { u32 lo = 0, hi = nchunks - 1; while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (loff[mid] <= t0) lo = mid; else hi = mid - 1; } c = lo; }
    u32 cend = loff[c + 1];
    for (u32 t = t0; t < t1; t++) {
        while (t >= cend) { c++; cend = loff[c + 1]; }
        const u64 e = mv[(u64)cstart[c] + (t - loff[c])];
        const u32 q = CH_REC_POS(e), delta = CH_REC_DELTA(e), size7 = CH_REC_SIZE(e);
        const bool alone = CH_REC_ALONE(e);
        long long x0 = (long long)q - 1 - CH_SHORT_CHAIN;
        const bool walk = delta != 0 || !alone || cx.SA != nullptr;     // (else only the chain's end matters: flags alone are read on the way)
        for (;;) {
            const long long x = x0 - (long long)lane;
            const u64 xc = x >= 0 ? (u64)x : 0ull;
            const u32 fl = cx.pinfo[xc];
            const u32 r = walk ? cx.rank[xc] : 0u;
            const long long xn = x0 - 64;                                 // the next step's first position: is the chain going on there?
            const bool goes_on = xn >= 0 && (cx.pinfo[xn >= 0 ? xn : 0] & 0x80u);
            const bool ok = x >= 0 && (fl & 0x80u);
            const u64 mk = __ballot(ok);
            const int cnt = mk == ~0ull ? 64 : __ffsll((unsigned long long)~mk) - 1;      // parked positions from the top of these 64
            const bool last = lane + 1 == cnt && !(cnt == 64 && goes_on);
            if (lane < cnt) {
                if (walk) cx.at((u32)xc, r, delta, size7, alone, last, CH_REC_EQB(e));
                else if (last && cx.out) cx.out[cx.rank[xc]] = cx.byte_before((u32)xc);
            }
            if (cnt < 64 || !goes_on) break;
            x0 -= 64;
        }
    }
}

// elements still tied over all chunks, and the exclusive sums of the chunks' kept-record counts for chunk_long_chains_kernel (one
// workgroup: at most a few 10^4 chunks)
__global__ __launch_bounds__(1024) void chunk_total_kernel(const u32 *__restrict__ ccount, u32 nchunks, unsigned long long *__restrict__ result,
                                                           const u32 *__restrict__ lcount, u32 *__restrict__ loff)
{
    __shared__ u64 sm[16];
    __shared__ u32 carry;
    u64 s = 0;
    for (u32 i = threadIdx.x; i < nchunks; i += 1024) s += ccount[i];
    s = wave_scan_inclusive(s, OpAdd());
    if (lane_id() == 63) sm[wave_id()] = s;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    if (threadIdx.x == 0) { u64 t = 0; for (int w = 0; w < 16; w++) t += sm[w]; result[CHS_TOTAL] = t; }
    if (!lcount) return;
    __shared__ u32 wtot[16];
    for (u32 i0 = 0; i0 < nchunks; i0 += 1024) {
        const u32 i = i0 + threadIdx.x;
        const u32 v = i < nchunks ? lcount[i] : 0u;
        const u32 inc = wave_scan_inclusive(v, OpAdd());
        __syncthreads();                                      // (wtot / carry of the previous step have been read)
        if (lane_id() == 63) wtot[wave_id()] = inc;
        __syncthreads();
        u32 before = carry;
        for (int w = 0; w < wave_id(); w++) before += wtot[w];
        if (i < nchunks) loff[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
    }
    __syncthreads();
    if (threadIdx.x == 0) loff[nchunks] = carry;
}

// When most of the list has settled (or parked) the chunks are nearly empty, and a round pays a workgroup's set-up per chunk for a handful
// of elements: the survivors are then copied, chunk after chunk (groups stay whole and in order), into the other pair of list arrays and
// new chunks are cut over the dense list.  off: exclusive sums of ccount.
__global__ __launch_bounds__(256) void chunk_compact_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ head, const u32 *__restrict__ cstart,
                                                            const u32 *__restrict__ ccount, const u32 *__restrict__ off, u32 *__restrict__ nidx, u32 *__restrict__ nhead)
{
    const u32 c = blockIdx.x;
    const u64 base = cstart[c], o = off[c];
    const u32 cnt = ccount[c];
    for (u32 i = threadIdx.x; i < cnt; i += 256) { nidx[o + i] = idx[base + i]; nhead[o + i] = head[base + i]; }
}

// chunks c0 .. c0 + nch over list slots [lo, hi): chunk i nominally starts at lo + i * S, actually at the first group start at
// or after that (groups here have at most CH_GROUP_MAX members; a wave looks at 64 slots at a time).  One wave per chunk; it also
// walks the chunk once to see whether a group of more than CH_CAP members is in it (-> cwide).
__device__ __forceinline__ u64 chunk_group_start_at_or_after(const u32 *__restrict__ head, u64 lo, u64 hi, u64 s, int lane)
{
    if (s > hi) s = hi;
    if (s <= lo || s >= hi) return s;
    for (;;) {
        const u64 i = s + (u64)lane;
        const bool st = i >= hi || head[i] != head[i - 1];
        const u64 m = __ballot(st);
        if (m) return s + (u64)(__ffsll((unsigned long long)m) - 1);
        s += 64;
    }
}
__global__ __launch_bounds__(256) void chunk_init_kernel(const u32 *__restrict__ head, u64 lo, u64 hi, u32 S, u32 c0, u32 nch,
                                                         u32 *__restrict__ cstart, u32 *__restrict__ ccount, u32 *__restrict__ mvcount, u8 *__restrict__ cwide,
                                                         bool may_be_wide)
{
    const u32 i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= nch) return;
    const u64 s = chunk_group_start_at_or_after(head, lo, hi, lo + (u64)i * S, lane);
    const u64 e = i + 1 < nch ? chunk_group_start_at_or_after(head, lo, hi, lo + (u64)(i + 1) * S, lane) : hi;
    bool wide = false;
    if (may_be_wide) {
        u64 last = s;                           // the most recent group start seen
        for (u64 b = s; b < e && !wide; b += 64) {
            const u64 q = b + (u64)lane;
            const bool st = q < e && (q == s || head[q] != head[q - 1]);
            const u64 m = __ballot(st);
            if (m) {
                if (b + (u64)(__ffsll((unsigned long long)m) - 1) - last > CH_CAP) wide = true;
                last = b + 63u - (u64)__clzll((long long)m);
            }
        }
        if (e - last > CH_CAP) wide = true;
    }
    if (lane == 0) {
        cstart[c0 + i] = (u32)s;
        ccount[c0 + i] = (u32)(e - s);
        mvcount[c0 + i] = 0;
        cwide[c0 + i] = wide ? 1 : 0;
    }
}

// what is left when no group splits any more (equal infinite words): the members take their group's slots in list order -- as records
// (delta = place in the group, alone, own byte and suffix-array slot written by the record pass), so that the chains parked behind
// them are laid out the same way
__global__ __launch_bounds__(256) void chunk_rest_records_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ head, const u32 *__restrict__ cstart,
                                                                 const u32 *__restrict__ ccount, u64 *__restrict__ mv, u32 *__restrict__ mvcount, bool parked)
{
    const u32 c = blockIdx.x;
    const u64 base = cstart[c];
    const u32 cnt = ccount[c];
    for (u32 i = threadIdx.x; i < cnt; i += 256) {
        const u32 hh = head[base + i];
        u32 o = 0;
        while (o < i && head[base + i - o - 1] == hh) o++;
        mv[base + i] = parked ? CH_REC(idx[base + i], o, 0u, true, 0u) | CH_REC_SELF : (((u64)(hh + o) << 32) | (u64)idx[base + i]);
    }
    if (threadIdx.x == 0) mvcount[c] = cnt;
}

// ---- the one-off order, by GROUP RECORDS ----------------------------------------------------------------------------------------
// The list only has to come out with its groups in the order of their smallest positions.  Sorting the elements for that (round 2, and
// the first form of this file) moves 32 bytes per element and pass; the groups are two or three members each on text, so one record per
// group -- (size << 32 | smallest position, list index of its first member) -- is sorted instead, the sizes are scanned in sorted order
// and the members copied to their places (go_expand_kernel).  Members of groups larger than CH_CAP are not ordered at all: they are
// compacted, in list (= SA) order, behind everything else and become the big list.
// tile_counts[t] (exclusive-scanned in place before go_write_kernel; entry [tiles] = totals): low word = group records, high = big elements
__global__ __launch_bounds__(DG_THREADS) void go_count_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ head, u64 a, u64 *__restrict__ tile_counts)
{
    __shared__ u32 hd[DG_SPAN];
    __shared__ u64 startm[DG_SPAN / 64];
    __shared__ u64 wsum[DG_THREADS / 64];
    const int tid = threadIdx.x;
    const long long e0 = (long long)blockIdx.x * DG_OWN - DG_CAP;
    DgSlots ds;
    dg_detect(idx, head, a, e0, hd, startm, ds);
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        const u32 sl = (u32)j * DG_THREADS + tid;
        if (ds.kind[j] == 1 && ds.gs[j] == sl) c += 1ull;
        if (ds.kind[j] == 2) c += 1ull << 32;
    }
    c = wave_scan_inclusive(c, OpAdd());
    if (lane_id() == 63) wsum[wave_id()] = c;
    __syncthreads();
    if (tid == 0) { u64 t = 0; for (int w = 0; w < DG_THREADS / 64; w++) t += wsum[w]; tile_counts[blockIdx.x] = t; }
}
__global__ __launch_bounds__(DG_THREADS) void go_write_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ head, u64 a, const u64 *__restrict__ tile_off, u64 tiles,
                                                              u64 *__restrict__ rkeys, u32 *__restrict__ rvals, u32 *__restrict__ st_idx, u32 *__restrict__ st_head,
                                                              bool pairs /* n <= 2^31: bit 31 of a position is free to tell the two record forms apart */)
{
    __shared__ u32 hd[DG_SPAN];
    __shared__ u32 pos[DG_SPAN];
    __shared__ u64 startm[DG_SPAN / 64];
    __shared__ u32 wcnt[2][DG_ITEMS][DG_THREADS / 64];           // per (item row, wave): records, big elements -- rows are consecutive slot ranges
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long e0 = (long long)blockIdx.x * DG_OWN - DG_CAP;
    DgSlots ds;
    dg_detect(idx, head, a, e0, hd, startm, ds);
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++)
        if (ds.kind[j] == 1) pos[j * DG_THREADS + tid] = ds.idx[j];
    u64 rm[DG_ITEMS], bm[DG_ITEMS];
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        const u32 sl = (u32)j * DG_THREADS + tid;
        rm[j] = __ballot(ds.kind[j] == 1 && ds.gs[j] == sl);
        bm[j] = __ballot(ds.kind[j] == 2);
        if (lane == 0) { wcnt[0][j][wv] = (u32)__popcll(rm[j]); wcnt[1][j][wv] = (u32)__popcll(bm[j]); }
    }
    __syncthreads();
    const u64 off = tile_off[blockIdx.x], tot = tile_off[tiles];
    const u64 a_small = a - (tot >> 32);                       // the larger groups' members land behind the smaller groups' (list order kept)
#pragma unroll
    for (int j = 0; j < DG_ITEMS; j++) {
        u32 rbefore = 0, bbefore = 0;
        for (int jj = 0; jj <= j; jj++)
            for (int w = 0; w < DG_THREADS / 64; w++)
                if (jj < j || w < wv) { rbefore += wcnt[0][jj][w]; bbefore += wcnt[1][jj][w]; }
        const u32 sl = (u32)j * DG_THREADS + tid;
        if (ds.kind[j] == 1 && ds.gs[j] == sl) {
            u32 mn = 0xffffffffu, mx = 0;
            for (u32 m = 0; m < ds.sz[j]; m++) { const u32 q = pos[sl + m]; mn = q < mn ? q : mn; mx = q > mx ? q : mx; }
            const u64 r = (u32)off + rbefore + (u32)__popcll(rm[j] & lanemask_lt());
            if (pairs && ds.sz[j] == 2) {
                // a group of two travels whole: (larger position << 32 | smaller position, head) -- its members need not be fetched again
                rkeys[r] = ((u64)mx << 32) | mn;
                rvals[r] = ds.h[j];
            } else {
                rkeys[r] = ((u64)((pairs ? 0x80000000u : 0u) | ds.h[j]) << 32) | mn;        // the group's head = the SA slot of its first member
                rvals[r] = ds.sz[j];
            }
        }
        if (ds.kind[j] == 2) {
            const u64 o = a_small + (off >> 32) + bbefore + (u32)__popcll(bm[j] & lanemask_lt());
            st_idx[o] = ds.idx[j];
            st_head[o] = ds.h[j];
        }
    }
}
// record forms (go_write_kernel): pairs == false: (head = SA slot of the first member << 32 | smallest position, size); pairs == true:
// bit 63 set: the same with the flag; bit 63 clear: a group of two, (larger position << 32 | smaller position, head)
// STATIC PARKING (see PARKED CHAINS at the top).  The records are sorted by smallest position, so the group of a record's successors, if it
// is a group at all, is the NEXT record: record r is linked when record r + 1 holds exactly its members' successors (same size, member for
// member one position on -- the members of a group stand in position order since round 0's stable sort -- and no member ends its factor).
// A record with GO_RUN_MIN linked records in a row from itself on lies in a long repeat, GO_RUN_MIN groups or more from its end: its
// members never enter the list; they are parked at once (flag + size in pinfo), behind the records nearer the repeat's end, which are
// refined (and park one by one, dynamically, once they are what leads the chain).  Short runs stay in the list whole: chains of a few
// positions cost more in the record pass than their groups cost in the two or three rounds that settle them.
#define GO_RUN_MIN 64
__global__ __launch_bounds__(256) void go_link_kernel(const u64 *__restrict__ rk, const u32 *__restrict__ rv, u64 groups, const u32 *__restrict__ SA, bool pairs,
                                                      const u32 *__restrict__ fstart, u32 k, u64 n, u64 *__restrict__ linkw)
{
    __shared__ u32 fs[CH_FS];
    for (u32 i = threadIdx.x; i < k; i += 256) fs[i] = fstart[i];
    __syncthreads();
    // is x the last position of its factor?  <=> x + 1 starts a factor, or x ends the text
    auto fend = [&](u32 x) -> bool {
        if ((u64)x + 1 >= n) return true;
        u32 lo = 0, hi = k;                       // first factor start > x
        while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (fs[mid] > x) hi = mid; else lo = mid + 1; }
        return lo < k && fs[lo] == x + 1u;
    };
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    bool link = false;
    if (r + 1 < groups) {
        const u64 k0 = rk[r], k1 = rk[r + 1];
        if (pairs && !(k0 >> 63)) {
            if (!(k1 >> 63)) {
                const u32 mn = (u32)k0, mx = (u32)(k0 >> 32);
                link = (u32)k1 == mn + 1u && (u32)(k1 >> 32) == mx + 1u && !fend(mn) && !fend(mx);
            }
        } else if (!pairs || (k1 >> 63)) {
            const u32 sz = rv[r];
            if (sz == rv[r + 1] && sz <= CH_PARK_MAX && (u32)k1 == (u32)k0 + 1u) {
                const u32 hm = pairs ? 0x7fffffffu : 0xffffffffu;
                const u64 h0 = (u32)(k0 >> 32) & hm, h1 = (u32)(k1 >> 32) & hm;
                link = true;
                for (u32 i = 0; i < sz; i++) {
                    const u32 a = SA[h0 + i];
                    if (SA[h1 + i] != a + 1u || fend(a)) { link = false; break; }
                }
            }
        }
    }
    const u64 m = __ballot(link);
    if (lane_id() == 0) linkw[r >> 6] = m;
}
// GO_RUN_MIN (= 64) link bits from record r on, all set?  (linkw: two zero words behind the last)
__device__ __forceinline__ bool go_parked(const u64 *__restrict__ linkw, u64 r)
{
    static_assert(GO_RUN_MIN == 64, "one funnel of two words");
    if (!linkw) return false;
    // (the 64-bit window at bit r of the stream, cut out with 32-bit funnel shifts: a 64-bit VALU shift by a variable amount is what the
    // gfx950 erratum of DESIGN.md section 10 bites -- it did, in one instantiation of the scan's reduce sweep, once that sweep evaluated
    // its functor in every lane: tools/check_shift64.py names it, the static-parking child failed on it)
    const u64 w = r >> 6;
    const u32 b = (u32)r & 63u;
    const u64 lo = linkw[w], hi = linkw[w + 1];
    const u32 a0 = (u32)lo, a1 = (u32)(lo >> 32), a2 = (u32)hi, a3 = (u32)(hi >> 32);
    const u32 s = b & 31u;
    const u32 x0 = b < 32u ? __builtin_amdgcn_alignbit(a1, a0, s) : __builtin_amdgcn_alignbit(a2, a1, s);
    const u32 x1 = b < 32u ? __builtin_amdgcn_alignbit(a2, a1, s) : __builtin_amdgcn_alignbit(a3, a2, s);
    return (x0 & x1) == 0xffffffffu;
}
struct GoSizeIn {
    const u64 *rk; const u32 *rv; bool pairs; const u64 *linkw;
    __device__ __forceinline__ u32 operator()(u64 j) const { return go_parked(linkw, j) ? 0u : ((pairs && !(rk[j] >> 63)) ? 2u : rv[j]); }
};
// elements that enter the list: the last record's offset + its size
__global__ void go_listed_kernel(const u32 *doff, GoSizeIn zin, u64 groups, u64 *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = groups ? (u64)doff[groups - 1] + (u64)zin(groups - 1) : 0ull;
}
// members of the sorted groups to their places: a pair comes out of its record; the others are read from the suffix array itself -- a
// group is the slot range [head, head + size) there, and all its members carry that head -- a lane copying its own group when it is
// short, the wave together the longer ones (one scattered read per group)
__global__ __launch_bounds__(256) void go_expand_kernel(const u64 *__restrict__ rkeys, const u32 *__restrict__ rvals, const u32 *__restrict__ doff, u64 groups,
                                                        const u32 *__restrict__ SA, u32 *__restrict__ st_idx, u32 *__restrict__ st_head,
                                                        bool pairs, u8 *__restrict__ pinfo /* null, or: every member's group size goes there (PARKED CHAINS) */,
                                                        const u64 *__restrict__ linkw /* null, or: go_link_kernel's bits (static parking) */)
{
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    const int lane = lane_id();
    u32 e = 0, sz = 0, d = 0;
    if (j < groups) {
        const u64 k = rkeys[j];
        const u32 v = rvals[j];
        d = doff[j];
        if (go_parked(linkw, j)) {
            // parked from the start: no list entry, the flag in position space (the loops below see a group of size 0)
            if (pairs && !(k >> 63)) { pinfo[(u32)k] = (u8)(2u | 0x80u); pinfo[(u32)(k >> 32)] = (u8)(2u | 0x80u); }
            else {
                const u32 hd = (u32)(k >> 32) & (pairs ? 0x7fffffffu : 0xffffffffu);
                for (u32 t = 0; t < v; t++) pinfo[SA[(u64)hd + t]] = (u8)(v | 0x80u);
            }
        } else if (pairs && !(k >> 63)) {
            st_idx[d] = (u32)k; st_idx[d + 1] = (u32)(k >> 32);
            st_head[d] = v; st_head[d + 1] = v;
            if (pinfo) { pinfo[(u32)k] = 2; pinfo[(u32)(k >> 32)] = 2; }
        } else { e = (u32)(k >> 32) & (pairs ? 0x7fffffffu : 0xffffffffu); sz = v; }
    }
    {
        // (a short group's members: four loads issued together, from the group's first slot where it has fewer -- not one behind a test each)
        const bool few = sz && sz <= 4;
        u32 pi[4];
#pragma unroll
        for (u32 t = 0; t < 4; t++) pi[t] = SA[(u64)e + (few && t < sz ? t : 0u)];
        if (few) {
#pragma unroll
            for (u32 t = 0; t < 4; t++) if (t < sz) { st_idx[d + t] = pi[t]; st_head[d + t] = e; if (pinfo) pinfo[pi[t]] = (u8)sz; }
        }
    }
    u64 longm = __ballot(sz > 4);
    while (longm) {
        const int r = __ffsll((unsigned long long)longm) - 1;
        longm &= longm - 1;
        const u32 re = shfl_t(e, r), rs = shfl_t(sz, r), rd = shfl_t(d, r);
        for (u32 t = (u32)lane; t < rs; t += 64) {
            const u32 pp = SA[(u64)re + t];
            st_idx[rd + t] = pp; st_head[rd + t] = re;
            if (pinfo) pinfo[pp] = (u8)(rs < 127u ? rs : 127u);
        }
    }
}

// ---- the big list ---------------------------------------------------------------------------------------------------------
struct BlIn {
    const u32 *head;
    __device__ __forceinline__ u32 operator()(u64 j) const { return (j == 0 || head[j] != head[j - 1]) ? 1u : 0u; }
};
template <bool CYCLIC>
struct BlOut {
    const u32 *head; const u32 *idx; int rb; const u32 *rank; u64 n; u64 h; const u32 *fstart; u64 k;
    u64 *bk; u32 *bv;
    u64 *k23, *k23_sort; u32 *j_sort;        // quadrupled step (null otherwise): see DgBigOut
    __device__ __forceinline__ void operator()(u64 j, u32 before) const
    {
        const u32 st = (j == 0 || head[j] != head[j - 1]) ? 1u : 0u;
        const u64 ord = (u64)before + st - 1;
        const u64 p = idx[j];
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
// after the sorts, in sorted order.  bl_flags_kernel looks at every element and its predecessor once -- first of its group (the ordinal
// changes), first of its new subgroup (any of the three ranks changes) -- and fetches the element's position; what follows reads
// the flag bytes and positions in sequence.  (The regrouping scan used to do the comparisons itself, in both of its sweeps: three
// gathers through the second sort's permutation per element and sweep.)
#define BLF_GROUP 1u
#define BLF_SUB   2u
__global__ __launch_bounds__(256) void bl_flags_kernel(const u64 *__restrict__ bk, const u32 *__restrict__ src, const u64 *__restrict__ k23,
                                                       const u32 *__restrict__ bv, u64 m, int rb, u8 *__restrict__ flags, u32 *__restrict__ t_idx)
{
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    const bool valid = j < m;
    const u64 key = valid ? bk[j] : 0ull;
    const u32 s = valid ? (src ? src[j] : (u32)j) : 0u;
    const u64 q = (valid && k23) ? k23[s] : 0ull;
    if (valid) t_idx[j] = bv[s];
    u64 kprev = shfl_up_t(key, 1), qprev = shfl_up_t(q, 1);
    if (lane_id() == 0 && valid && j > 0) { kprev = bk[j - 1]; qprev = k23 ? k23[src ? src[j - 1] : (u32)(j - 1)] : 0ull; }
    if (!valid) return;
    const bool gstart = j == 0 || (kprev >> rb) != (key >> rb);
    const bool sstart = gstart || kprev != key || qprev != q;
    flags[j] = (u8)((gstart ? BLF_GROUP : 0u) | (sstart ? BLF_SUB : 0u));
}
// regrouping scan (max on both halves, see OpMax2): high word = 1 + index of the element's group start, low word = 1 + index of its
// subgroup start
struct BlFlagIn {
    const u8 *flags;
    __device__ __forceinline__ u64 operator()(u64 j) const
    {
        const u32 f = flags[j];
        return ((u64)((f & BLF_GROUP) ? (u32)j + 1u : 0u) << 32) | (u64)((f & BLF_SUB) ? (u32)j + 1u : 0u);
    }
};
// new heads, final bytes, new ranks (every gather of the round is done by now); and, for the split that follows, every element's
// subgroup start and -- written by the subgroup's last element -- the subgroup's size
struct BlRegroupOut {
    const u8 *flags; const u32 *oldhead; u64 m;
    const u32 *t_idx; u32 *t_head; u32 *rank; PrevSym prev; u8 *out; unsigned long long *result;
    u32 *rstart; u32 *rsize; bool all_bytes;
    __device__ __forceinline__ void operator()(u64 j, u64 v) const       // inclusive (max, max) scan value
    {
        const u32 gidx = (u32)(v >> 32) - 1u, sidx = (u32)v - 1u;
        const u32 newhead = oldhead[j] + (sidx - gidx);        // sorting keeps every group on its own slots, all holding its old head
        const u32 p = t_idx[j];
        t_head[j] = newhead;
        if (sidx != gidx) rank[p] = newhead;
        const bool last_of_sub = j + 1 == m || (flags[j + 1] & BLF_SUB);
        // (with parked chains: the byte of every member at its place in its group's slots, alone or not -- ChainCtx::at)
        if (out && (all_bytes || (sidx == (u32)j && last_of_sub))) out[newhead + ((u32)j - sidx)] = prev(p);
        rstart[j] = sidx;
        if (last_of_sub) rsize[sidx] = (u32)j - sidx + 1u;
        const u64 splitm = __ballot(sidx != gidx);
        if (splitm && lane_id() == __ffsll((unsigned long long)__ballot(true)) - 1 &&
            __hip_atomic_load(&result[CHS_SPLIT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            __hip_atomic_store(&result[CHS_SPLIT], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
// class of every element of the (regrouped) big list: 0 alone (finished), 1 in a group of 2 .. CH_GROUP_MAX (leaves for a chunk),
// 2 stays.  Group sizes from the run lengths of equal heads: an inclusive max-scan gives every element its run's first index, the
// run's last element writes the length there.
struct BlRunIn {
    const u32 *head;
    __device__ __forceinline__ u32 operator()(u64 i) const { return (i == 0 || head[i] != head[i - 1]) ? (u32)i + 1u : 0u; }
};
struct BlRunOut {
    const u32 *head; u64 m; u32 *rstart; u32 *rsize;
    __device__ __forceinline__ void operator()(u64 i, u32 v) const
    {
        const u32 s = v - 1u;
        rstart[i] = s;
        if (i + 1 == m || head[i + 1] != head[i]) rsize[s] = (u32)i - s + 1u;
    }
};
__device__ __forceinline__ u32 bl_class(const u32 *__restrict__ rstart, const u32 *__restrict__ rsize, u64 i)
{
    const u32 sz = rsize[rstart[i]];
    return sz > CH_GROUP_MAX ? 2u : (sz >= 2 ? 1u : 0u);
}
struct BlSplitIn {
    const u32 *rstart; const u32 *rsize;
    __device__ __forceinline__ u64 operator()(u64 i) const { const u32 c = bl_class(rstart, rsize, i); return (u64)(c == 1 ? 1u : 0u) | ((u64)(c == 2 ? 1u : 0u) << 32); }
};
struct BlSplitOut {
    const u32 *rstart; const u32 *rsize; const u32 *t_idx; const u32 *t_head; u64 m;
    u32 *x_idx, *x_head;          // where the leaving elements go (the chunk store's tail)
    u32 *s_idx, *s_head;          // the next big list
    unsigned long long *result;
    __device__ __forceinline__ void operator()(u64 i, u64 before) const
    {
        const u32 c = bl_class(rstart, rsize, i);
        if (c == 1) { const u32 o = (u32)before; x_idx[o] = t_idx[i]; x_head[o] = t_head[i]; }
        else if (c == 2) {
            const u32 o = (u32)(before >> 32); s_idx[o] = t_idx[i]; s_head[o] = t_head[i];
            // (the next round's group ordinals are below this count: it sizes the sort key)
            const u64 firsts = __ballot(rstart[i] == (u32)i);
            if (lane_id() == __ffsll((unsigned long long)firsts) - 1) atomicAdd(&result[CHS_BIGGROUPS], (unsigned long long)__popcll(firsts));
        }
        if (i + 1 == m) { result[CHS_EXIT] = (u64)(u32)before + (c == 1 ? 1u : 0u); result[CHS_STAY] = (before >> 32) + (c == 2 ? 1u : 0u); }
    }
};

// (BWTS_ROUND_TRACE only) which of the big list's groups hold one value of the rank at h / of all three ranks: such a group cannot split
__global__ void bl_diag_flag_kernel(const u64 *__restrict__ bk0, const u64 *__restrict__ k23, u64 m, int rb, u8 *__restrict__ f1, u8 *__restrict__ f23)
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 || j >= m) return;
    const u64 a = bk0[j], b = bk0[j - 1];
    if ((a >> rb) != (b >> rb)) return;
    if (a != b) f1[a >> rb] = 1;
    if (k23 && k23[j] != k23[j - 1]) f23[a >> rb] = 1;
}
__global__ void bl_diag_count_kernel(const u64 *__restrict__ bk0, u64 m, int rb, const u8 *__restrict__ f1, const u8 *__restrict__ f23, unsigned long long *cnt)
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = j < m;
    const u64 o = in ? bk0[j] >> rb : 0;
    const u64 c1 = __ballot(in && !f1[o]), c3 = __ballot(in && !f1[o] && !f23[o]);
    if (lane_id() == 0) { if (c1) atomicAdd(&cnt[0], (unsigned long long)__popcll(c1)); if (c3) atomicAdd(&cnt[1], (unsigned long long)__popcll(c3)); }
    if (in && j + 1 == m) cnt[2] = o + 1;
}

// (BWTS_ROUND_TRACE only) elements of the regrouped big list by log2 of their group's size
__global__ void bl_diag_sizes_kernel(const u32 *__restrict__ rstart, const u32 *__restrict__ rsize, u64 m, unsigned long long *hist)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || rstart[i] != (u32)i) return;
    const u32 sz = rsize[i];
    atomicAdd(&hist[31 - __clz(sz)], (unsigned long long)sz);
    atomicAdd(&hist[32 + 31 - __clz(sz)], 1ull);
}
static void bl_diag_sizes(bwts_ctx *ctx, const u32 *rstart, const u32 *rsize, u64 m, const char *what)
{
    unsigned long long *d = nullptr, hh[64];
    if (hipMalloc((void **)&d, sizeof(hh)) != hipSuccess) return;
    (void)hipMemsetAsync(d, 0, sizeof(hh), ctx->stream);
    bl_diag_sizes_kernel<<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream>>>(rstart, rsize, m, d);
    (void)hipMemcpyAsync(hh, d, sizeof(hh), hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    fprintf(stderr, "[chunks] %s: elements (groups) by log2 group size:", what);
    for (int b = 0; b < 32; b++) if (hh[b]) fprintf(stderr, " %d:%llu(%llu)", b, hh[b], hh[32 + b]);
    fprintf(stderr, "\n");
}

// (BWTS_ROUND_TRACE only) elements of the chunks by the size of their group: 2, 3-4, 5-16, 17-64, 65-256, 257-2048
__global__ __launch_bounds__(256) void chunk_diag_sizes_kernel(const u32 *__restrict__ head, const u32 *__restrict__ cstart, const u32 *__restrict__ ccount, unsigned long long *hist)
{
    const u32 c = blockIdx.x, cnt = ccount[c];
    const u64 base = cstart[c];
    for (u32 i = threadIdx.x; i < cnt; i += 256) {
        if (i && head[base + i] == head[base + i - 1]) continue;
        u32 e = i + 1;
        while (e < cnt && head[base + e] == head[base + i]) e++;
        const u32 sz = e - i;
        const int cls = sz <= 2 ? 0 : sz <= 4 ? 1 : sz <= 16 ? 2 : sz <= 64 ? 3 : sz <= 256 ? 4 : 5;
        atomicAdd(&hist[cls], (unsigned long long)sz);
    }
}
static void chunk_diag_sizes(bwts_ctx *ctx, const u32 *head, const u32 *cstart, const u32 *ccount, u32 nchunks, u32 round)
{
    unsigned long long *d = nullptr, hh[6];
    if (!nchunks || hipMalloc((void **)&d, sizeof(hh)) != hipSuccess) return;
    (void)hipMemsetAsync(d, 0, sizeof(hh), ctx->stream);
    chunk_diag_sizes_kernel<<<dim3(nchunks), dim3(256), 0, ctx->stream>>>(head, cstart, ccount, d);
    (void)hipMemcpyAsync(hh, d, sizeof(hh), hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    fprintf(stderr, "[chunks] before round %u, chunk elements by group size: 2: %llu  3-4: %llu  5-16: %llu  17-64: %llu  65-256: %llu  257-2048: %llu\n", round, hh[0], hh[1], hh[2], hh[3], hh[4], hh[5]);
}

static u32 chunk_nominal_size(u64 a)
{
    // about 16 K chunks, between one and eight tiles each
    u64 s = (a / 16384 + 1023) / 1024 * 1024;
    if (s < CH_TILE) s = CH_TILE;
    if (s > 8 * CH_TILE) s = 8 * CH_TILE;
    return (u32)s;
}

// Same contract as dense_rounds().  *handled = false: this form does not apply (short list, no room) and nothing was changed.
template <bool CYCLIC>
static int chunk_rounds(bwts_ctx *ctx, const u8 *d_T, u64 n, const Alphabet &al, const u32 *d_fstart, u64 k, SortSpace &sp,
                        ActiveList cur, u64 a0, u32 *SA, bool need_sa, u32 *rounds_io, bool *handled)
{
    *handled = false;
    if (a0 > 0xffffffffull || a0 < CH_MIN_LIST) return BWTS_OK;
    const bool round_trace = [ctx] { const char *e = bwts_knob(ctx, "BWTS_ROUND_TRACE"); return e && atoi(e) == 1; }();
#define CH_TRY(call) do { const int rc__ = (call); if (rc__ != BWTS_OK) { if (round_trace) fprintf(stderr, "[chunks] line %d: rc %d\n", __LINE__, rc__); return rc__; } } while (0)
#define CH_HIP(call) do { const hipError_t e__ = (call); if (e__ != hipSuccess) { ctx->last_hip = (int)e__; if (round_trace) fprintf(stderr, "[chunks] line %d: hip error %d\n", __LINE__, (int)e__); return BWTS_E_HIP; } } while (0)
#define CH_FAIL(why) do { if (round_trace) fprintf(stderr, "[chunks] invariant: %s (round %u)\n", why, rounds); return BWTS_E_INTERNAL; } while (0)
    const size_t e4 = align_up((size_t)a0 * 4, 256), e8 = align_up((size_t)a0 * 8, 256);
    u32 S = chunk_nominal_size(a0);
    const u64 maxchunks = a0 / S + 1024;                 // every append adds at most one ragged chunk; rounds are capped at 80
    const size_t ct4 = align_up((size_t)(maxchunks + 1) * 4, 256);
    char *base = nullptr, *ob = nullptr;
    const size_t ct1 = align_up((size_t)(maxchunks + 1), 256);
    // parked chains (see the top of this file): cyclic sort with the factors in the round kernel's LDS; BWTS_PARK=0 switches them off
    // (opt-in: measured, it does not pay on this chip -- profiles/history/r04_parked_chains.md -- and stays as a tested alternate path)
    const bool parking = CYCLIC && k <= CH_FS && [ctx] { const char *e = bwts_knob(ctx, "BWTS_PARK"); return e && atoi(e) == 1; }();
    const bool static_parking = parking && [ctx] { const char *e = bwts_knob(ctx, "BWTS_PARK_STATIC"); return e && atoi(e) == 1; }();
    const size_t pin_bytes = parking ? align_up((size_t)n, 256) : 0;
    int rc = aux_reserve(ctx, 4 * e4 + e8 + 5 * ct4 + ct1 + pin_bytes, &base);
    if (rc == BWTS_E_NOMEM) return BWTS_OK;
    CH_TRY(rc);
    rc = aux_reserve_slot(ctx, 1, 2 * e8 + 2 * e4, &ob);
    if (rc == BWTS_E_NOMEM) return BWTS_OK;
    CH_TRY(rc);
    *handled = true;
    u32 *st_idx = (u32 *)base, *st_head = (u32 *)(base + e4);
    u64 *mv = (u64 *)(base + 2 * e4);
    u32 *cstart = (u32 *)(base + 2 * e4 + e8), *ccount = (u32 *)(base + 2 * e4 + e8 + ct4), *mvcount = (u32 *)(base + 2 * e4 + e8 + 2 * ct4);
    u8 *cwide = (u8 *)(base + 2 * e4 + e8 + 3 * ct4);
    u32 *lcount = (u32 *)(base + 2 * e4 + e8 + 3 * ct4 + ct1), *loff = (u32 *)(base + 2 * e4 + e8 + 4 * ct4 + ct1);
    u8 *pinfo = parking ? (u8 *)(base + 2 * e4 + e8 + 5 * ct4 + ct1) : nullptr;
    u32 *alt_idx = (u32 *)(base + 2 * e4 + e8 + 5 * ct4 + ct1 + pin_bytes), *alt_head = (u32 *)(base + 3 * e4 + e8 + 5 * ct4 + ct1 + pin_bytes);   // (chunk_compact_kernel)
    if (parking) CH_HIP(hipMemsetAsync(pinfo, 0, (size_t)n, ctx->stream));
    u64 *slots = ctx->d_small + SM_CHSLOT;
    const int rb = CYCLIC ? bitlen_u64(n - 1) : bitlen_u64(n);
    PrevSym prev{sp.carry_src, d_T, n, d_fstart, k};
    u8 *out = CYCLIC ? sp.carry_out : nullptr;
    u32 rounds = *rounds_io;

    // ---- one-off order: group records sorted by the group's smallest position, members copied to their places; the larger groups'
    // members compacted behind them (see go_write_kernel) ----
    u64 a_small = 0, a_listed = 0;           // elements of the smaller groups; those of them that enter the list (the others are parked at once)
    {
        const u64 tiles = (a0 + DG_OWN - 1) / DG_OWN;
        const bool pairs = n <= 0x80000000ull;                        // (positions and list indices below 2^31)
        const u64 gmax = a0 / 2 + 1;                                  // a group has at least two members
        const size_t g8 = align_up((size_t)gmax * 8, 256), g4 = align_up((size_t)gmax * 4, 256);
        // records and their sort buffers inside the order block (2 x 8 + 3 x 4 bytes per group <= 14 bytes per element), tile counts behind
        u64 *rk[2] = {(u64 *)ob, (u64 *)(ob + g8)};
        u32 *rv[2] = {(u32 *)(ob + 2 * g8), (u32 *)(ob + 2 * g8 + g4)};
        u32 *doff = (u32 *)(ob + 2 * g8 + 2 * g4);
        u64 *tcount = (u64 *)(ob + 2 * g8 + 3 * g4);
        const size_t tc_bytes = align_up((size_t)(tiles + 1) * 8, 256);
        const u64 link_words = (gmax + 255) / 256 * 4 + 2;              // go_link_kernel's bits: a word per wave, two zero words behind
        u64 *linkw = parking ? (u64 *)(ob + 2 * g8 + 3 * g4 + tc_bytes) : nullptr;
        if (2 * g8 + 3 * g4 + tc_bytes + (parking ? align_up((size_t)link_words * 8, 256) : 0) > 2 * e8 + 2 * e4) CH_FAIL("order block too small");
        {
            SpanGuard g(ctx, BWTS_K_RERANK, a0, 14 * a0);
            go_count_kernel<<<dim3((unsigned)tiles), dim3(DG_THREADS), 0, ctx->stream>>>(cur.idx, cur.head, a0, tcount);
            CH_HIP(hipMemsetAsync(tcount + tiles, 0, sizeof(u64), ctx->stream));
            ScanLoadArr<u64> tin{tcount};
            ScanStoreArr<u64> tout{tcount};
            CH_TRY((device_scan<false, u64>(ctx, tiles + 1, tin, tout, OpAdd(), (u64)0, sp.scan_temp)));
            CH_HIP(hipMemcpyAsync(slots + CHS_TOTAL, tcount + tiles, sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
            go_write_kernel<<<dim3((unsigned)tiles), dim3(DG_THREADS), 0, ctx->stream>>>(cur.idx, cur.head, a0, tcount, tiles, rk[0], rv[0], st_idx, st_head, pairs);
            CH_HIP(hipGetLastError());
        }
        CH_TRY(read_small(ctx, SM_CHSLOT, CH_SLOT_WORDS));
        const u64 tot = ctx->h_small[SM_CHSLOT + CHS_TOTAL];
        const u64 groups = (u32)tot, bigs = tot >> 32;
        if (bigs > a0 || groups > gmax) CH_FAIL("group records");
        a_small = a0 - bigs;
        if (groups) {
            SortPlan op;
            op.keys[0] = rk[0]; op.keys[1] = rk[1];
            op.vals[0] = rv[0]; op.vals[1] = rv[1];
            op.tile_hist = sp.tile_hist; op.scan_temp = sp.scan_temp;
            int ores = 0;
            const int pb = bitlen_u64(n - 1);
            CH_TRY(radix_sort_pairs(ctx, op, groups, pb < 1 ? 1 : pb, &ores));
            SpanGuard g(ctx, BWTS_K_RERANK, a_small, 20 * a_small);
            const unsigned gblocks = (unsigned)((groups + 255) / 256);
            if (!static_parking) linkw = nullptr;
            if (static_parking) {
                // static parking: records deep inside long repeats never enter the list (go_link_kernel)
                CH_HIP(hipMemsetAsync(linkw + (u64)gblocks * 4, 0, 2 * sizeof(u64), ctx->stream));
                go_link_kernel<<<dim3(gblocks), dim3(256), 0, ctx->stream>>>(rk[ores], rv[ores], groups, SA, pairs, d_fstart, (u32)k, n, linkw);
            }
            GoSizeIn zin{rk[ores], rv[ores], pairs, linkw};
            ScanStoreArr<u32> zout{doff};
            CH_TRY((device_scan<false, u32>(ctx, groups, zin, zout, OpAdd(), 0u, sp.scan_temp)));
            go_expand_kernel<<<dim3(gblocks), dim3(256), 0, ctx->stream>>>(rk[ores], rv[ores], doff, groups, SA, st_idx, st_head, pairs, pinfo, linkw);
            CH_HIP(hipGetLastError());
            if (static_parking) {
                // how much of the smaller groups did enter the list: the last record's offset + its size
                go_listed_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(doff, zin, groups, slots + CHS_TOTAL);
                CH_TRY(read_small(ctx, SM_CHSLOT, CH_SLOT_WORDS));
                a_listed = ctx->h_small[SM_CHSLOT + CHS_TOTAL];
                if (a_listed > a_small) CH_FAIL("listed elements");
            } else a_listed = a_small;
        }
    }
    u64 m_big = a0 - a_small;
    if (round_trace) fprintf(stderr, "[chunks] list %llu: in chunks %llu (parked at once: %llu; nominal chunk %u), big list %llu\n", (unsigned long long)a0,
                             (unsigned long long)a_listed, (unsigned long long)(a_small - a_listed), S, (unsigned long long)m_big);

    // ---- big list buffers (the order sort's block, free again) ----
    const size_t m4 = align_up((size_t)m_big * 4, 256), m8 = align_up((size_t)m_big * 8, 256);
    const bool step4_ok = [ctx] { const char *e = bwts_knob(ctx, "BWTS_DENSE_STEP"); return !(e && atoi(e) == 2); }();
    const int nk = step4_ok ? 3 : 1;
    u32 *bl_idx[2] = {nullptr, nullptr}, *bl_head[2] = {nullptr, nullptr}, *t_idx = nullptr, *t_head = nullptr, *bv[2] = {nullptr, nullptr}, *sv1 = nullptr;
    u64 *bk[2] = {nullptr, nullptr}, *k23 = nullptr, *sk1 = nullptr;
    u8 *bflags = nullptr;
    u64 big_groups = 0;                     // groups in the big list: the ordinals of a round's sort key lie below it
    int blc = 0;
    if (m_big) {
        char *bb = nullptr;
        // (nothing outside this function's own buffers has been written so far: without room for the big list the tile form takes over,
        // as it does when the first two blocks do not fit)
        const bool deny = [ctx] { const char *e = bwts_knob(ctx, "BWTS_BIGLIST_NOMEM"); return e && atoi(e) == 1; }();      // (test switch)
        rc = deny ? BWTS_E_NOMEM : aux_reserve_slot(ctx, 1, 4 * m8 + 9 * m4 + align_up((size_t)m_big, 256), &bb);
        if (rc == BWTS_E_NOMEM) {
            if (round_trace) fprintf(stderr, "[chunks] no room for the big list (%llu elements): the tile form takes over\n", (unsigned long long)m_big);
            *handled = false;
            return BWTS_OK;
        }
        CH_TRY(rc);
        char *q = bb;
        for (int i = 0; i < 2; i++) { bl_idx[i] = (u32 *)q; q += m4; bl_head[i] = (u32 *)q; q += m4; }
        t_idx = (u32 *)q; q += m4; t_head = (u32 *)q; q += m4;
        bv[0] = (u32 *)q; q += m4; bv[1] = (u32 *)q; q += m4; sv1 = (u32 *)q; q += m4;
        bk[0] = (u64 *)q; q += m8; bk[1] = (u64 *)q; q += m8; k23 = (u64 *)q; q += m8; sk1 = (u64 *)q; q += m8;
        bflags = (u8 *)q;
        CH_HIP(hipMemcpyAsync(bl_idx[0], st_idx + a_small, m_big * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
        CH_HIP(hipMemcpyAsync(bl_head[0], st_head + a_small, m_big * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
    }
    // ---- chunks over the smaller groups ----
    u32 nchunks = 0;
    // (the larger groups' members, which go_write_kernel left at [a_small, a0), are in the big list's own buffers by now: the store
    // behind the listed elements is free, chunks leaving the big list are appended there)
    if (a_listed < a_small) S = chunk_nominal_size(a_listed);
    u64 tail = a_listed;
    if (a_listed) {
        nchunks = (u32)((a_listed + S - 1) / S);
        chunk_init_kernel<<<dim3((nchunks + 3) / 4), dim3(256), 0, ctx->stream>>>(st_head, 0, a_listed, S, 0, nchunks, cstart, ccount, mvcount, cwide, false);
        CH_HIP(hipGetLastError());
    }

    u64 a_chunks = a_listed;                // elements in chunks after the last evaluated round
    bool wide_possible = false;             // some chunk may be flagged WIDE: the second instantiation is launched as well
    if (m_big) {
        // the one-off order left every group of more than CH_CAP members in the big list; those of up to CH_GROUP_MAX go to chunks right
        // away (unordered, like everything that leaves the big list later): their chunks are flagged WIDE and take the sorting instantiation of the round kernel
        CH_HIP(hipMemsetAsync(slots, 0, CH_SLOT_WORDS * sizeof(u64), ctx->stream));
        {
            SpanGuard g(ctx, BWTS_K_RERANK, m_big, 28 * m_big);
            BlRunIn nin{bl_head[0]};
            BlRunOut nout{bl_head[0], m_big, bv[0], bv[1]};
            CH_TRY((device_scan<true, u32>(ctx, m_big, nin, nout, OpMax(), 0u, sp.scan_temp)));
            if (round_trace) bl_diag_sizes(ctx, bv[0], bv[1], m_big, "groups of more than 256 after round 0");
            BlSplitIn sin{bv[0], bv[1]};
            BlSplitOut sout{bv[0], bv[1], bl_idx[0], bl_head[0], m_big, st_idx + tail, st_head + tail, bl_idx[1], bl_head[1], (unsigned long long *)slots};
            CH_TRY((device_scan<false, u64>(ctx, m_big, sin, sout, OpAdd(), (u64)0, sp.scan_temp)));
        }
        CH_TRY(read_small(ctx, SM_CHSLOT, CH_SLOT_WORDS));
        const u64 m_exit = ctx->h_small[SM_CHSLOT + CHS_EXIT], m_stay = ctx->h_small[SM_CHSLOT + CHS_STAY];
        big_groups = ctx->h_small[SM_CHSLOT + CHS_BIGGROUPS];
        if (m_exit + m_stay != m_big || tail + m_exit > a0 || big_groups * (CH_GROUP_MAX + 1) > m_stay) CH_FAIL("first split of the big list");
        if (m_exit) {
            const u32 add = (u32)((m_exit + S - 1) / S);
            if ((u64)nchunks + add > maxchunks) CH_FAIL("chunk table full");
            chunk_init_kernel<<<dim3((add + 3) / 4), dim3(256), 0, ctx->stream>>>(st_head, tail, tail + m_exit, S, nchunks, add, cstart, ccount, mvcount, cwide, true);
            CH_HIP(hipGetLastError());
            nchunks += add;
            tail += m_exit;
            a_chunks += m_exit;
            wide_possible = true;
        }
        blc = 1;
        m_big = m_stay;
        if (round_trace) fprintf(stderr, "[chunks] groups of up to %d members leave the big list at once: %llu elements, %llu stay\n", (int)CH_GROUP_MAX,
                                 (unsigned long long)m_exit, (unsigned long long)m_stay);
    }

    u64 h = (u64)al.hstep;
    const int hshift = nk == 3 ? 2 : 1;
    bool finished = false, stable = false;
    // (static parking has taken the long repeats out; what the first list round meets are short ones, settled sooner by refining them
    // than by chains of a few positions: groups park dynamically from the second list round on)
    u32 list_rounds = 0;
    while (!finished) {
#ifdef CH_PROFILE
        const int B = 1;
#else
        const int B = m_big ? 1 : 2;        // rounds per host round trip
#endif
        CH_HIP(hipMemsetAsync(slots, 0, CH_SLOTS * CH_SLOT_WORDS * sizeof(u64), ctx->stream));
        u64 hs[CH_SLOTS];
        for (int b = 0; b < B; b++) {
            unsigned long long *res = (unsigned long long *)(slots + b * CH_SLOT_WORDS);
            hs[b] = h;
            if (round_trace && nchunks) chunk_diag_sizes(ctx, st_head, cstart, ccount, nchunks, rounds + 1);
            if (nchunks) {
                SpanGuard g(ctx, BWTS_K_ROUND, 0, 0);          // (elements and bytes are added below, once the round's true size is known)
#define CH_LAUNCH_P(NK, FS, WD, PK) chunk_round_kernel<CYCLIC, NK, FS, WD, PK><<<dim3(nchunks), dim3(CH_THREADS), 0, ctx->stream>>>(st_idx, st_head, cstart, ccount, cwide, mv, mvcount, \
                                                                                                        sp.rank, n, h, d_fstart, k, prev, out, res, pinfo, (!static_parking || list_rounds > 0) ? 1u : 0u)
#define CH_LAUNCH(NK, FS, WD) do { if (CYCLIC && FS && parking) CH_LAUNCH_P(NK, (CYCLIC && FS), WD, (CYCLIC && FS)); else CH_LAUNCH_P(NK, FS, WD, false); } while (0)
                const bool fsl = CYCLIC && k <= CH_FS;
                if (nk == 3) { if (fsl) CH_LAUNCH(3, CYCLIC, false); else CH_LAUNCH(3, false, false); }
                else { if (fsl) CH_LAUNCH(1, CYCLIC, false); else CH_LAUNCH(1, false, false); }
                if (wide_possible) {
                    // (behind the other one: a chunk whose last large group has just split is taken over in the NEXT round)
                    if (nk == 3) { if (fsl) CH_LAUNCH(3, CYCLIC, true); else CH_LAUNCH(3, false, true); }
                    else { if (fsl) CH_LAUNCH(1, CYCLIC, true); else CH_LAUNCH(1, false, true); }
                }
#undef CH_LAUNCH
#undef CH_LAUNCH_P
                CH_HIP(hipGetLastError());
            }
            if (m_big) {
                // the big list's gathers read the same version of the ranks as the chunks': before the moves are applied
                SpanGuard g(ctx, BWTS_K_RERANK, m_big, 30 * m_big);
                BlIn fin{bl_head[blc]};
                BlOut<CYCLIC> fout{bl_head[blc], bl_idx[blc], rb, sp.rank, n, h, d_fstart, k, bk[0], bv[0],
                                   nk == 3 ? k23 : nullptr, nk == 3 ? bk[1] : nullptr, nk == 3 ? bv[1] : nullptr};
                CH_TRY((device_scan<false, u32>(ctx, m_big, fin, fout, OpAdd(), 0u, sp.scan_temp)));
                if (round_trace) {
                    const u64 gcap = m_big / (CH_GROUP_MAX + 1) + 2;
                    u8 *df = nullptr; unsigned long long hc[3] = {0, 0, 0};
                    if (hipMalloc((void **)&df, 2 * gcap + 32) == hipSuccess) {
                        (void)hipMemsetAsync(df, 0, 2 * gcap + 32, ctx->stream);
                        unsigned long long *dc = (unsigned long long *)(df + ((2 * gcap + 7) & ~7ull));
                        const unsigned gb = (unsigned)((m_big + 255) / 256);
                        bl_diag_flag_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(bk[0], nk == 3 ? k23 : nullptr, m_big, rb, df, df + gcap);
                        bl_diag_count_kernel<<<dim3(gb), dim3(256), 0, ctx->stream>>>(bk[0], m_big, rb, df, df + gcap, dc);
                        (void)hipMemcpyAsync(hc, dc, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream);
                        (void)hipStreamSynchronize(ctx->stream);
                        (void)hipFree(df);
                        fprintf(stderr, "[chunks] big list at h %llu: %llu elements in %llu groups; in groups with one rank at h: %llu, with one rank triple (cannot split): %llu\n",
                                (unsigned long long)h, (unsigned long long)m_big, hc[2], hc[0], hc[1]);
                    }
                }
            }
            if (nchunks) {
                SpanGuard g(ctx, BWTS_K_ROUND, 0, 0);
                const ChainCtx cx{sp.rank, pinfo, prev, out, nullptr};
                if (parking) chunk_apply_records_kernel<2><<<dim3(nchunks), dim3(256), 0, ctx->stream>>>(mv, cstart, mvcount, cx, lcount);
                else chunk_apply_records_kernel<0><<<dim3(nchunks), dim3(256), 0, ctx->stream>>>(mv, cstart, mvcount, cx, lcount);
                chunk_total_kernel<<<dim3(1), dim3(1024), 0, ctx->stream>>>(ccount, nchunks, res, parking ? lcount : nullptr, loff);
                if (parking) chunk_long_chains_kernel<<<dim3(2048), dim3(256), 0, ctx->stream>>>(mv, cstart, loff, nchunks, cx);
                CH_HIP(hipGetLastError());
            }
            if (m_big) {
                const int big_bits = (big_groups > 1 ? bitlen_u64(big_groups - 1) : 1) + rb;          // ordinals < big_groups
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
                    CH_TRY(radix_sort_pairs(ctx, bp, m_big, 2 * rb, &r1));
                    u64 *kin = bp.keys[r1], *kout = bp.keys[r1 ^ 1];
                    u32 *vin = bp.vals[r1], *vout = bp.vals[r1 ^ 1];
                    {
                        SpanGuard g(ctx, BWTS_K_RERANK, m_big, 20 * m_big);
                        dg_stage2_keys_kernel<<<dim3((unsigned)((m_big + 255) / 256)), dim3(256), 0, ctx->stream>>>(vin, bk[0], m_big, kin);
                        CH_HIP(hipGetLastError());
                    }
                    bp.keys[0] = kin; bp.keys[1] = kout;
                    bp.vals[0] = vin; bp.vals[1] = vout;
                    CH_TRY(radix_sort_pairs(ctx, bp, m_big, big_bits, &rbig));
                    sorted_k1 = bp.keys[rbig]; src = bp.vals[rbig]; positions = bv[0];
                } else {
                    bp.keys[0] = bk[0]; bp.keys[1] = bk[1];
                    bp.vals[0] = bv[0]; bp.vals[1] = bv[1];
                    CH_TRY(radix_sort_pairs(ctx, bp, m_big, big_bits, &rbig));
                    sorted_k1 = bk[rbig]; positions = bv[rbig];
                }
                SpanGuard g(ctx, BWTS_K_RERANK, m_big, 60 * m_big);
                bl_flags_kernel<<<dim3((unsigned)((m_big + 255) / 256)), dim3(256), 0, ctx->stream>>>(sorted_k1, src, nk == 3 ? k23 : nullptr, positions, m_big, rb, bflags, t_idx);
                CH_HIP(hipGetLastError());
                // (the sort buffers are free again: subgroup starts and sizes go there)
                BlFlagIn rin{bflags};
                BlRegroupOut rout{bflags, bl_head[blc], m_big, t_idx, t_head, sp.rank, prev, out, res, bv[0], bv[1], parking};
                CH_TRY((device_scan<true, u64>(ctx, m_big, rin, rout, OpMax2(), (u64)0, sp.scan_temp)));
                if (round_trace) bl_diag_sizes(ctx, bv[0], bv[1], m_big, "big list regrouped");
                BlSplitIn sin{bv[0], bv[1]};
                BlSplitOut sout{bv[0], bv[1], t_idx, t_head, m_big, st_idx + tail, st_head + tail, bl_idx[blc ^ 1], bl_head[blc ^ 1], res};
                CH_TRY((device_scan<false, u64>(ctx, m_big, sin, sout, OpAdd(), (u64)0, sp.scan_temp)));
            }
            h = h > (1ull << 60) ? h : h << hshift;
            list_rounds++;
        }
        CH_TRY(read_small(ctx, SM_CHSLOT, CH_SLOTS * CH_SLOT_WORDS));
        for (int b = 0; b < B; b++) {
            const u64 *r = ctx->h_small + SM_CHSLOT + b * CH_SLOT_WORDS;
            // 32 algorithmic bytes per list element and round: position + head in, three successor ranks, position + head out, rank update
            if (nchunks) { ctx->tm.k[BWTS_K_ROUND].elems += a_chunks; ctx->tm.k[BWTS_K_ROUND].alg_bytes += 32 * a_chunks; }
            if (finished) continue;          // (a round enqueued behind the last one: it ran, over what was left, and changed nothing)
            rounds++;
            if (r[CHS_ERR]) {
                if (round_trace) fprintf(stderr, "[chunks] chunk %llu of %u: group [%d, %d) plen %llu len %llu rp %llu slot %llu\n", (unsigned long long)r[5] - 1, nchunks,
                                         (int)(r[6] >> 32), (int)(u32)r[6], (unsigned long long)(r[7] >> 48), (unsigned long long)((r[7] >> 32) & 0xffff),
                                         (unsigned long long)((r[7] >> 16) & 0xffff), (unsigned long long)(r[7] & 0xffff));
                CH_FAIL("a chunk met a group larger than it may hold");
            }
            const u64 in_chunks = nchunks ? r[CHS_TOTAL] : 0;
            u64 m_exit = 0, m_stay = 0;
            if (m_big) {
                m_exit = r[CHS_EXIT]; m_stay = r[CHS_STAY];
                big_groups = r[CHS_BIGGROUPS];
                if (m_exit + m_stay > m_big || tail + m_exit > a0 || big_groups * (CH_GROUP_MAX + 1) > m_stay) CH_FAIL("big list split counts");
            }
            if (in_chunks > a_chunks) CH_FAIL("chunks grew");
#ifdef CH_PROFILE
            if (round_trace && b == 0) {
                (void)hipMemcpy(ctx->h_small + SM_CHSLOT + 64, ctx->d_small + SM_CHSLOT + 64, 64 * 8 * sizeof(u64), hipMemcpyDeviceToHost);
                double ph[6] = {0, 0, 0, 0, 0, 0}, tot = 0;
                for (int q = 0; q < 64; q++) for (int i = 0; i < 6; i++) ph[i] += (double)ctx->h_small[SM_CHSLOT + 64 + 8 * q + i];
                for (int i = 0; i < 6; i++) tot += ph[i];
                fprintf(stderr, "[chunks] phase shares: load %.1f%% detect %.1f%% gather %.1f%% count %.1f%% emit/moves %.1f%% write %.1f%%  (cycles per element %.2f)\n", 100 * ph[0] / tot, 100 * ph[1] / tot,
                        100 * ph[2] / tot, 100 * ph[3] / tot, 100 * ph[4] / tot, 100 * ph[5] / tot, tot / (double)(a_chunks ? a_chunks : 1));
                (void)hipMemset(ctx->d_small + SM_CHSLOT + 64, 0, 64 * 8 * sizeof(u64));
            }
#endif
            if (round_trace) fprintf(stderr, "[chunks] round %u h %llu: chunks %llu -> %llu (parked %llu), big list %llu -> stays %llu, leaves %llu\n", rounds,
                                     (unsigned long long)hs[b], (unsigned long long)a_chunks, (unsigned long long)in_chunks, (unsigned long long)r[CHS_PARKED], (unsigned long long)m_big, (unsigned long long)m_stay, (unsigned long long)m_exit);
            if (m_exit) {
                const u32 add = (u32)((m_exit + S - 1) / S);
                if ((u64)nchunks + add > maxchunks) CH_FAIL("chunk table full");
                chunk_init_kernel<<<dim3((add + 3) / 4), dim3(256), 0, ctx->stream>>>(st_head, tail, tail + m_exit, S, nchunks, add, cstart, ccount, mvcount, cwide, true);
                wide_possible = true;
                CH_HIP(hipGetLastError());
                nchunks += add;
                tail += m_exit;
            }
            if (m_big) { blc ^= 1; m_big = m_stay; }
            a_chunks = in_chunks + m_exit;
            const u64 left = a_chunks + m_big;
            if (CYCLIC && rounds - 1 < BWTS_MAX_ROUND_STATS) ctx->tm.round_active[rounds - 1] = left;
            if (left == 0) { finished = true; continue; }
            // no group split: the partition is stable under doubling -- what is left are groups of equal infinite words
            if (CYCLIC && r[CHS_SPLIT] == 0) { finished = true; stable = true; continue; }
            if (!CYCLIC && hs[b] >= n) CH_FAIL("suffixes still tied at h >= n");      // suffixes are distinct; cannot happen
            if (rounds > 80) CH_FAIL("more than 80 rounds");
        }
        if (!finished && nchunks >= 64 && a_chunks > 0 && a_chunks * 3 < tail) {
            // the chunks are two thirds empty: a dense list, new chunks (chunk_compact_kernel)
            SpanGuard g(ctx, BWTS_K_ROUND, 0, 0);
            chunk_total_kernel<<<dim3(1), dim3(1024), 0, ctx->stream>>>(ccount, nchunks, (unsigned long long *)(slots + 3 * CH_SLOT_WORDS), ccount, loff);
            chunk_compact_kernel<<<dim3(nchunks), dim3(256), 0, ctx->stream>>>(st_idx, st_head, cstart, ccount, loff, alt_idx, alt_head);
            { u32 *t = st_idx; st_idx = alt_idx; alt_idx = t; t = st_head; st_head = alt_head; alt_head = t; }
            S = chunk_nominal_size(a_chunks);
            const u32 nc = (u32)((a_chunks + S - 1) / S);
            if (round_trace) fprintf(stderr, "[chunks] list compacted: %llu elements of %llu slots, %u chunks -> %u (nominal chunk %u)\n", (unsigned long long)a_chunks,
                                     (unsigned long long)tail, nchunks, nc, S);
            nchunks = nc;
            chunk_init_kernel<<<dim3((nchunks + 3) / 4), dim3(256), 0, ctx->stream>>>(st_head, 0, a_chunks, S, 0, nchunks, cstart, ccount, mvcount, cwide, wide_possible);
            CH_HIP(hipGetLastError());
            tail = a_chunks;
        }
    }
    if (need_sa) {
        SpanGuard g(ctx, BWTS_K_RERANK, n, 8 * n);
        u64 blocks = (n + 255) / 256; if (blocks > 16384) blocks = 16384;
        sa_from_rank_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(sp.rank, n, SA);
        CH_HIP(hipGetLastError());
    }
    if (stable) {
        // (a round enqueued behind the stable one has split nothing either: the lists are what they were)
        SpanGuard g(ctx, BWTS_K_EMIT, a_chunks + m_big, 10 * (a_chunks + m_big));
        if (nchunks && a_chunks) {
            const ChainCtx cx{sp.rank, pinfo, prev, out, need_sa ? SA : nullptr};
            chunk_rest_records_kernel<<<dim3(nchunks), dim3(256), 0, ctx->stream>>>(st_idx, st_head, cstart, ccount, mv, mvcount, parking);
            if (parking) chunk_apply_records_kernel<2><<<dim3(nchunks), dim3(256), 0, ctx->stream>>>(mv, cstart, mvcount, cx, lcount);
            else chunk_apply_records_kernel<1><<<dim3(nchunks), dim3(256), 0, ctx->stream>>>(mv, cstart, mvcount, cx, lcount);
            if (parking) {
                chunk_total_kernel<<<dim3(1), dim3(1024), 0, ctx->stream>>>(ccount, nchunks, (unsigned long long *)slots, lcount, loff);
                chunk_long_chains_kernel<<<dim3(2048), dim3(256), 0, ctx->stream>>>(mv, cstart, loff, nchunks, cx);
            }
            CH_HIP(hipGetLastError());
        }
        if (m_big) {
            DgRestIn rin{bl_head[blc]};
            DgRestOut rout{bl_idx[blc], bl_head[blc], prev, out, need_sa ? SA : nullptr};
            CH_TRY((device_scan<true, u32>(ctx, m_big, rin, rout, OpMax(), 0u, sp.scan_temp)));
        }
    }
    if (parking && out && !prev.P) {
        chunk_factor_heads_kernel<<<dim3((unsigned)((k + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_T, n, d_fstart, k, sp.rank, out);
        CH_HIP(hipGetLastError());
    }
    *rounds_io = rounds;
    return BWTS_OK;
#undef CH_FAIL
#undef CH_TRY
#undef CH_HIP
}
