// wide_path.h -- forward transform of inputs beyond the 32-bit index range (n > 2^32).  Included by forward.hip.
//
// The reference's indices are int / saidx_t (mk_bwts_sa.c:26-27, unbwts.c:12-13): it stops below 2^31.  The engine's main
// path keeps everything about a position in 32 bits and 36 n bytes of HBM, which ends at n = 2^32.  Here positions and ranks
// are 64-bit, and the sort is blocked so that its working set does not grow with n:
//   * the round-0 keys are never materialised for the whole text: a SEGMENT of positions at a time (keybuild0*_kernel with a
//     position offset), as often as somebody needs them;
//   * the Lyndon factors come from the same candidate search as on the main path, run segment by segment over the global
//     prefix minima of the tile minima;
//   * the positions are cut into BUCKETS by the top bits of their key -- a bucket is a contiguous range of the final order.  A
//     histogram of the key prefixes (per segment) gives every bucket's size and where each segment's share lands in it;
//     bucket by bucket the members are collected (stable), sorted by key (the same LSD passes; the position's bits above 32
//     ride in the byte stream that otherwise carries the emission byte), and finished: global rank = bucket base + group
//     head into the n-entry u64 rank array, output byte written, tied elements appended to the tied list;
//   * the tied list (a few thousand elements for i.i.d. data, 70 % of the positions for text: it lives in blocks taken as it
//     grows) goes through rounds of (group, rank of the h-th cyclic successor) sorting with 64-bit ranks until no group splits,
//     a part of whole groups at a time, in place.
// Memory: rank array 8 n, one segment of keys, one bucket's sort buffers: ~160 GiB at n = 12 GiB, against ~430 GiB for the
// main path's layout.  The inverse's 64-bit form is wide_inverse.h (dispatched from inverse_device_impl).
#define WIDE_PREFIX_BITS 12
#define WIDE_PREFIXES    (1u << WIDE_PREFIX_BITS)
#define WIDE_MAX_PARTS   4096u

__device__ __forceinline__ u64 factor_of64(const u64 *__restrict__ fstart, u64 k, u64 p)
{
    u64 lo = 0, hi = k - 1;
    while (lo < hi) {
        const u64 mid = (lo + hi + 1) >> 1;
        if (fstart[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}
struct PrevSym64 {       // T[cprev(p)] (mk_bwts_sa.c:172-188) through the factor list
    const u8 *T; u64 n; const u64 *fstart; u64 k;
    __device__ __forceinline__ u8 operator()(u64 p) const
    {
        const u64 f = factor_of64(fstart, k, p);
        if (fstart[f] == p) return T[(f + 1 < k ? fstart[f + 1] : n) - 1];
        return T[p - 1];
    }
};
__device__ __forceinline__ u64 cyclic_successor64(const u64 *__restrict__ fstart, u64 k, u64 n, u64 p, u64 h)
{
    const u64 f = factor_of64(fstart, k, p);
    const u64 s = fstart[f], L = (f + 1 < k ? fstart[f + 1] : n) - s;
    return cyclic_successor(p, s, L, h);
}

// keys of the positions close to a factor's end wrap around inside the factor (cyclic_patch[_vl]_kernel of the main path),
// restricted to the segment [pos0, lim) whose keys are in `keys`
__global__ __launch_bounds__(256) void cyclic_patch_wide_kernel(const u8 *__restrict__ T, u64 n, const u8 *__restrict__ codes, int bits, int msym,
                                                                const u64 *__restrict__ vtab /* variable-length codes, or null */, int key_bits,
                                                                int span, const u64 *__restrict__ fstart, u64 k, u64 *__restrict__ keys, u64 pos0, u64 lim)
{
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (span <= 0 || t >= k * (u64)span) return;
    const u64 f = t / (u64)span, j = t % (u64)span;
    const u64 s = fstart[f], e = f + 1 < k ? fstart[f + 1] : n;
    if (j >= e - s) return;
    const u64 p = e - 1 - j;
    if (p < pos0 || p >= lim) return;
    keys[p - pos0] = vtab ? vl_key_cyclic(T, vtab, key_bits, p, s, e) : cyclic_key(T, codes, bits, msym, p, s, e);
    // MI355X erratum (tools/check_shift64.py): a 64-bit shift must not take its amount from the wave's last
    // allocated VGPR.  This kernel compiled to 16 VGPRs with the key loop's shift amount in v15 and produced wrong keys whenever
    // waves shared a SIMD; one more allocated register behind it keeps the amount away from the end of the allocation.
    asm volatile("; keep v16 allocated" ::: "v16");
}

// per-segment histogram of the key prefixes
__global__ __launch_bounds__(256) void wide_prefix_hist_kernel(const u64 *__restrict__ keys, u64 count, int shift, u32 *__restrict__ hist)
{
    __shared__ u32 bins[WIDE_PREFIXES];
    for (u32 i = threadIdx.x; i < WIDE_PREFIXES; i += 256) bins[i] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += (u64)gridDim.x * 256) atomicAdd(&bins[(u32)(keys[i] >> shift)], 1u);
    __syncthreads();
    for (u32 i = threadIdx.x; i < WIDE_PREFIXES; i += 256) if (bins[i]) atomicAdd(&hist[i], bins[i]);
}

// members of bucket [plo, phi) among a segment's keys, in position order (stable compaction)
struct WideFilterIn {
    const u64 *keys; int shift; u32 plo, phi;
    __device__ __forceinline__ u32 operator()(u64 i) const { const u32 pf = (u32)(keys[i] >> shift); return pf >= plo && pf < phi ? 1u : 0u; }
};
// The byte stream beside (key, position low word) carries the position's bits above 32 -- or, when the key leaves its top
// four bits free above the bytes the passes sort on (key_bits <= 56: everything but the 64-bit keys of text), those bits ride in the key (no pass sorts on them)
// and the stream carries the output byte T[cprev(p)], which the key build leaves beside the segment's keys: the finish kernel then has
// no random read left.
#define WIDE_HI_SHIFT 60
struct WideFilterOut {
    const u64 *keys; int shift; u32 plo, phi; u64 pos0; u64 *bk; u32 *bv; u8 *bs;      // bk/bv/bs already point at this segment's share
    const u8 *segprev;          // the segment's output bytes (key build + wide_head_prev_kernel), or null: the stream carries position bits
    __device__ __forceinline__ void operator()(u64 i, u32 before) const
    {
        const u64 key = keys[i];
        const u32 pf = (u32)(key >> shift);
        if (pf >= plo && pf < phi) {
            const u64 p = pos0 + i;
            bv[before] = (u32)p;
            if (segprev) { bk[before] = key | ((p >> 32) << WIDE_HI_SHIFT); bs[before] = segprev[i]; }
            else { bk[before] = key; bs[before] = (u8)(p >> 32); }
        }
    }
};
// the output byte of a factor's first position is the factor's last byte (mk_bwts_sa.c:172-188); the key build wrote T[p - 1]
__global__ __launch_bounds__(256) void wide_head_prev_kernel(const u8 *__restrict__ T, u64 n, const u64 *__restrict__ fstart, u64 k, u8 *__restrict__ segprev,
                                                            u64 pos0, u64 lim)
{
    const u64 f = (u64)blockIdx.x * 256 + threadIdx.x;
    if (f >= k) return;
    const u64 s = fstart[f];
    if (s >= pos0 && s < lim) segprev[s - pos0] = T[(f + 1 < k ? fstart[f + 1] : n) - 1];
}

// The tied list: (position, group head) pairs in group order, held in blocks of 2^lg pairs that are taken from the device as the
// list grows (a text of 6 GiB leaves 4.5 * 10^9 positions tied after round 0; i.i.d. data a few thousand) -- positions in a
// block's first half, heads in its second.  The rounds rewrite it in place.
#define WIDE_TB_MAX 64
struct TiedList {
    u64 *const *tab; int lg;
    __device__ __forceinline__ u64 &pos(u64 i) const { return tab[i >> lg][i & ((1ull << lg) - 1ull)]; }
    __device__ __forceinline__ u64 &head(u64 i) const { return tab[i >> lg][(1ull << lg) + (i & ((1ull << lg) - 1ull))]; }
};

// a sorted bucket is finished: ranks, output bytes, tied elements (appended from list slot tbase on: the host has made room)
template <bool CARRY>
__global__ __launch_bounds__(256) void wide_bucket_finish_kernel(const u64 *__restrict__ K, const u32 *__restrict__ V, const u8 *__restrict__ S, u64 m, u64 base,
                                                                 const u64 *__restrict__ headw, const u64 *__restrict__ keepw, const u64 *__restrict__ pre,
                                                                 u64 *__restrict__ rank64, PrevSym64 prev, u8 *__restrict__ out,
                                                                 u64 tbase, u64 tied_cap, TiedList tl, u64 *__restrict__ overflow, u64 n)
{
    const int lane = lane_id();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < ((m + 63) & ~63ull); i += (u64)gridDim.x * 256) {
        if (i >= m) continue;
        const u64 w = i >> 6;
        const u64 hm = headw[w], km = keepw[w], pr = pre[w];
        const u64 below = lane == 63 ? hm : hm & ((2ull << lane) - 1ull);
        const u64 hloc = below ? (w << 6) + (u64)(63 - __clzll((long long)below)) : (pr >> 32);
        const u64 p = (u64)V[i] | ((CARRY ? K[i] >> WIDE_HI_SHIFT : (u64)S[i]) << 32);
        const u64 r = base + hloc;
        // (a position rebuilt from sorted data: should key build and prefix histogram ever disagree again, this is an error code,
        // not a write through a stale position -- the GPU fault of round 2's fuzz run)
        if (p >= n) { *overflow = 2; continue; }
        if (rank64) rank64[p] = r;
        out[base + i] = CARRY ? S[i] : prev(p);
        if ((km >> lane) & 1ull) {
            const u64 t = tbase + (u64)(u32)pr + (u64)__popcll(km & lanemask_lt());
            if (t < tied_cap) { tl.pos(t) = p; tl.head(t) = r; }
            else *overflow = 1;
        }
    }
}
__global__ void wide_add_tied_kernel(const u64 *__restrict__ keepw, const u64 *__restrict__ pre, u64 words, u64 *__restrict__ tied_count)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *tied_count += (u64)(u32)pre[words - 1] + (u64)__popcll(keepw[words - 1]);
}

// ---- rounds over the tied list (64-bit ranks) -------------------------------------------------------------------------
// A round takes the list in PARTS: whole groups, as many as the sort buffers hold.  A part gathers all its keys, sorts, regroups,
// and only then writes the new ranks of its elements; the next part starts after that.  So whoever reads the ranks of a group's
// members sees the group either entirely as the round found it or entirely refined, never a mixture -- which is all the
// comparison of two successors' ranks needs (a refined group's ranks lie inside the old group's range and order its members
// correctly; an old group's are all equal: "tied so far").  A mixture would not do: with heads as ranks, a member already
// showing head + 1 beside a larger group-mate still showing head orders the wrong way round -- the reason the main path's round
// kernels, whose workgroups run side by side, only read the rank array and apply the moves afterwards.  This is the order of
// work of Larsson & Sadakane's sequential sort, a part for a group.  A round in which no group splits has changed no rank, so it
// still proves that what remains are equal infinite words.  Survivors go back to the front of the list behind those of the
// parts before.
__global__ __launch_bounds__(64) void wide_cuts_kernel(TiedList tl, u64 a, u64 P, u64 *__restrict__ cuts, u32 max_parts)
{
    // cuts[0] = number of parts (or ~0: a group larger than a part, or more parts than the table holds), cuts[1 + i] = end of part i
    const int lane = threadIdx.x;
    u64 lo = 0;
    u32 np = 0;
    while (lo < a) {
        u64 e = a - lo > P ? lo + P : a;
        if (e < a) {
            // the last group start in (lo, e]
            u64 top = e, found = 0;
            while (top > lo) {
                const u64 i = top - (u64)lane;
                const bool st = top >= (u64)lane && i > lo && tl.head(i) != tl.head(i - 1);
                const u64 m = __ballot(st);
                if (m) { found = top - (u64)(__ffsll((unsigned long long)m) - 1); break; }
                top = top - lo > 64 ? top - 64 : lo;
            }
            if (!found) { if (lane == 0) cuts[0] = ~0ull; return; }
            e = found;
        }
        if (np >= max_parts) { if (lane == 0) cuts[0] = ~0ull; return; }
        if (lane == 0) cuts[1 + np] = e;
        np++;
        lo = e;
    }
    if (lane == 0) cuts[0] = np;
}
struct WideOrdIn {
    TiedList tl; u64 lo;
    __device__ __forceinline__ u32 operator()(u64 i) const { return (i == 0 || tl.head(lo + i) != tl.head(lo + i - 1)) ? 1u : 0u; }
};
struct WideOrdOut {
    TiedList tl; u64 lo; const u64 *rank64; u64 n; u64 h; const u64 *fstart; u64 k; int rb; u64 *bk; u32 *bv; u64 m; u64 *groups;
    __device__ __forceinline__ void operator()(u64 i, u32 before) const
    {
        const u32 st = (i == 0 || tl.head(lo + i) != tl.head(lo + i - 1)) ? 1u : 0u;
        const u64 ord = (u64)before + st - 1;
        const u64 r2 = rank64[cyclic_successor64(fstart, k, n, tl.pos(lo + i), h)];
        bk[i] = (ord << rb) | r2;
        bv[i] = (u32)i;
        if (i + 1 == m) *groups = (u64)before + st;
    }
};
struct WideRegroupOut {
    const u64 *bk; const u32 *bv; u64 m; TiedList tl; u64 lo; u64 *npos; u64 *nhead; u8 *state; PrevSym64 prev; u8 *out; u64 *split;
    __device__ __forceinline__ void operator()(u64 j, u64 v) const       // inclusive scan value of DgRegroupIn
    {
        const u32 gidx = (u32)(v >> 32) - 1u, sidx = (u32)v - 1u;
        const u64 kj = bk[j];
        const bool alone = sidx == (u32)j && (j + 1 == m || bk[j + 1] != kj);
        const u64 li = lo + (u64)bv[j];
        const u64 p = tl.pos(li);
        const u64 nh = tl.head(li) + (u64)(sidx - gidx);
        npos[j] = p; nhead[j] = nh;
        state[j] = (u8)((alone ? DG_DONE : DG_KEEP) | (sidx != gidx ? DG_MOVED : 0));
        if (alone) out[nh] = prev(p);
        const u64 sm = __ballot(sidx != gidx);
        if (sm && lane_id() == __ffsll((unsigned long long)__ballot(true)) - 1 &&
            __hip_atomic_load(split, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            __hip_atomic_store(split, (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
struct WideKeepIn {
    const u8 *state;
    __device__ __forceinline__ u32 operator()(u64 i) const { return (state[i] & 3) == DG_KEEP ? 1u : 0u; }
};
struct WideKeepOut {
    const u8 *state; const u64 *npos; const u64 *nhead; u64 *rank64; TiedList tl; const u64 *wp; u64 m; u64 *count;
    __device__ __forceinline__ void operator()(u64 j, u32 before) const
    {
        const u32 st = state[j];
        const bool keep = (st & 3) == DG_KEEP;
        if (st & DG_MOVED) rank64[npos[j]] = nhead[j];
        if (keep) { const u64 o = *wp + (u64)before; tl.pos(o) = npos[j]; tl.head(o) = nhead[j]; }
        if (j + 1 == m) *count = (u64)before + (keep ? 1u : 0u);
    }
};
__global__ void wide_advance_kernel(u64 *__restrict__ wp, const u64 *__restrict__ count)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *wp += *count;
}
struct WideRestIn {
    TiedList tl; u64 lo;
    __device__ __forceinline__ u32 operator()(u64 i) const { return (i == 0 || tl.head(lo + i) != tl.head(lo + i - 1)) ? (u32)i + 1u : 0u; }
};
struct WideRestOut {
    TiedList tl; u64 lo; PrevSym64 prev; u8 *out;
    __device__ __forceinline__ void operator()(u64 i, u32 v) const { out[tl.head(lo + i) + ((u64)i - (u64)(v - 1u))] = prev(tl.pos(lo + i)); }
};

// ---- few ties: the groups are ordered by comparing their members' rotations in the text itself ------------------------------
// i.i.d.-like inputs leave a few thousand positions tied after the first sort, in pairs and triples that differ a few symbols on.
// For them the n-entry rank array -- 8 n bytes, one random 8-byte write per position -- is only ever read at a few thousand
// places, so the forward first runs WITHOUT it: the buckets are finished without rank writes, and every tied group is put in
// order by one thread comparing the members' cyclic rotations (each inside its own Lyndon factor: mk_bwts_sa.c:74-112 orders
// positions by their factor's rotation, read round and round) from the first symbol the keys did not cover.  Anything that does
// not fit this form -- more than WIDE_DIRECT_MAX tied positions, a group of more than WIDE_DIRECT_GROUP members, two rotations
// that agree beyond WIDE_DIRECT_DEPTH symbols (long repeats, equal long factors) -- makes the caller run again with the rank array.
#define WIDE_DIRECT_MAX   (1ull << 22)
#define WIDE_DIRECT_GROUP 32
#define WIDE_DIRECT_DEPTH 4096ull
// sampled keys of one segment, sorted: how many neighbours are equal?  (Two in 2^20 already mean millions of ties in 2^33 positions.)
__global__ __launch_bounds__(256) void wide_sample_keys_kernel(const u64 *__restrict__ keys, u64 count, u32 samples, u64 *__restrict__ out, u32 *__restrict__ vals)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < samples) { out[i] = keys[(u64)i * (count / samples)]; vals[i] = i; }
}
__global__ __launch_bounds__(256) void wide_sample_ties_kernel(const u64 *__restrict__ sorted, u32 samples, u64 *__restrict__ ties)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const bool eq = i + 1 < samples && sorted[i] == sorted[i + 1];
    const u64 m = __ballot(eq);
    if (m && lane_id() == __ffsll((unsigned long long)m) - 1) atomicAdd((unsigned long long *)ties, (unsigned long long)__popcll(m));
}
struct WideGroupIn {
    TiedList tl;
    __device__ __forceinline__ u32 operator()(u64 i) const { return (i == 0 || tl.head(i) != tl.head(i - 1)) ? 1u : 0u; }
};
struct WideGroupOut {
    TiedList tl; u32 *gstart; u64 a; u64 *groups;
    __device__ __forceinline__ void operator()(u64 i, u32 before) const
    {
        const bool st = i == 0 || tl.head(i) != tl.head(i - 1);
        if (st) gstart[before] = (u32)i;
        if (i + 1 == a) *groups = (u64)before + (st ? 1u : 0u);
    }
};
__global__ __launch_bounds__(64) void wide_direct_groups_kernel(TiedList tl, const u32 *__restrict__ gstart, u64 ngroups, u64 a, const u8 *__restrict__ T, u64 n,
                                                                const u64 *__restrict__ fstart, u64 k, u64 skip, PrevSym64 prev, u8 *__restrict__ out,
                                                                u64 *__restrict__ unresolved)
{
    const u64 g = (u64)blockIdx.x * 64 + threadIdx.x;
    if (g >= ngroups) return;
    const u64 s = gstart[g], e = g + 1 < ngroups ? (u64)gstart[g + 1] : a;
    const u32 cnt = (u32)(e - s);
    if (cnt > WIDE_DIRECT_GROUP) { *unresolved = 1; return; }
    u64 pos[WIDE_DIRECT_GROUP], fs[WIDE_DIRECT_GROUP], fl[WIDE_DIRECT_GROUP];
    for (u32 i = 0; i < cnt; i++) {
        const u64 p = tl.pos(s + i);
        const u64 f = factor_of64(fstart, k, p);
        pos[i] = p; fs[i] = fstart[f]; fl[i] = (f + 1 < k ? fstart[f + 1] : n) - fs[i];
    }
    // insertion sort; less(x, y): the rotations at pos[x], pos[y], from symbol `skip` on
    bool bad = false;
    for (u32 i = 1; i < cnt && !bad; i++) {
        const u64 p = pos[i], ps = fs[i], pl = fl[i];
        u32 j = i;
        while (j > 0) {
            const u64 q = pos[j - 1], qs = fs[j - 1], ql = fl[j - 1];
            u64 op = (p - ps + skip) % pl, oq = (q - qs + skip) % ql;
            bool p_less = false, decided = false;
            // two periodic words that agree on (sum of their periods) symbols are equal for ever (Fine and Wilf): such members are
            // identical rotations of identical factors, emit the same byte, and may stay in either order
            const u64 lim = pl + ql < WIDE_DIRECT_DEPTH ? pl + ql : WIDE_DIRECT_DEPTH;
            for (u64 d = 0; d < lim; d++) {
                const u8 cp = T[ps + op], cq = T[qs + oq];
                if (cp != cq) { p_less = cp < cq; decided = true; break; }
                if (++op == pl) op = 0;
                if (++oq == ql) oq = 0;
            }
            if (!decided && pl + ql > WIDE_DIRECT_DEPTH) { bad = true; break; }
            if (!p_less) break;
            pos[j] = q; fs[j] = qs; fl[j] = ql;
            j--;
        }
        pos[j] = p; fs[j] = ps; fl[j] = pl;
    }
    if (bad) { *unresolved = 1; return; }
    const u64 h0 = tl.head(s);
    for (u32 i = 0; i < cnt; i++) out[h0 + i] = prev(pos[i]);
}

// room for `elements` list entries; the block table goes to the device whenever it has changed (or `push` asks)
static int wide_tied_ensure(bwts_ctx *ctx, u64 elements, int lg, u64 **d_tab, bool push)
{
    if (!ctx->tied_blk.empty() && lg > ctx->tied_blk_lg) {
        // (blocks of an earlier, smaller call: nothing in them is live)
        HIPC(hipStreamSynchronize(ctx->stream));
        for (char *b : ctx->tied_blk) HIPC(hipFree(b));
        ctx->tied_blk.clear();
    }
    if (ctx->tied_blk.empty()) { ctx->tied_blk.reserve(WIDE_TB_MAX); if (lg > ctx->tied_blk_lg) ctx->tied_blk_lg = lg; }
    const int L = ctx->tied_blk_lg;
    const u64 want = (elements + (1ull << L) - 1ull) >> L;
    if (want > WIDE_TB_MAX) return BWTS_E_NOMEM;
    while (ctx->tied_blk.size() < want) {
        const double t0 = wall_ms();
        void *p = nullptr;
        if (hipMalloc(&p, (size_t)16 << L) != hipSuccess) { (void)hipGetLastError(); return BWTS_E_NOMEM; }
        ctx->tied_blk.push_back((char *)p);
        ctx->host_ms[BWTS_H_ARENA_ALLOC] += wall_ms() - t0;
        push = true;
    }
    if (push && !ctx->tied_blk.empty())
        HIPC(hipMemcpyAsync(d_tab, ctx->tied_blk.data(), ctx->tied_blk.size() * sizeof(char *), hipMemcpyHostToDevice, ctx->stream));
    return BWTS_OK;
}

struct WideKnobs { int seg_log2; u64 bucket_cap; u64 part; int tblock_log2; int direct; };
static WideKnobs wide_knobs(const bwts_ctx *ctx)
{
    WideKnobs kn{30, 1ull << 30, 0, 0, -1};
    if (const char *e = bwts_knob(ctx, "BWTS_WIDE_DIRECT")) kn.direct = atoi(e) ? 1 : 0;     // 1: ties by direct comparison only, 0: rank array at once
    if (const char *e = bwts_knob(ctx, "BWTS_WIDE_PART")) { const long long v = atoll(e); if (v >= 64) kn.part = (u64)v; }             // elements per part of a round
    if (const char *e = bwts_knob(ctx, "BWTS_WIDE_TBLOCK_LOG2")) { const int v = atoi(e); if (v >= 6 && v <= 30) kn.tblock_log2 = v; } // tied-list block size
    if (const char *e = bwts_knob(ctx, "BWTS_WIDE_SEG_LOG2")) { const int v = atoi(e); if (v >= 11 && v <= 31) kn.seg_log2 = v; }      // >= log2(KB_TILE)
    if (const char *e = bwts_knob(ctx, "BWTS_WIDE_BUCKET")) { const long long v = atoll(e); if (v >= 256 && v <= 0xff000000ll) kn.bucket_cap = (u64)v; }
    return kn;
}

// direct: without the rank array (see wide_direct_groups_kernel); *again is set when the input turns out to need it
static int forward_wide_run(bwts_ctx *ctx, const u8 *d_T, u64 n, u8 *d_out, bool direct, bool *again)
{
    *again = false;
    if (n > (1ull << 36)) return BWTS_E_RANGE;
    WideKnobs kn = wide_knobs(ctx);
    // larger buckets (fewer collection passes over the text) while rank array + bucket buffers + in/out leave room: 12 GiB of DNA
    // take 2.03 s with buckets of 2^30 elements (before the carried byte), 1.86 s with 2^31, 1.62 s with 3 * 2^30 (203 GiB on the device)
    if (!bwts_knob(ctx, "BWTS_WIDE_BUCKET")) {
        // (without the rank array there is room for the largest bucket the 32-bit sort indices allow: 4 instead of 5 passes over the
        // text at 12 GiB, 7 instead of 9 at 24 GiB)
        if (direct && n <= (48ull << 30)) kn.bucket_cap = 0xff000000ull;
        else if (n <= (13ull << 30)) kn.bucket_cap = 3ull << 30;
        else if (n <= (14ull << 30) || direct) kn.bucket_cap = 1ull << 31;
    }
    const u64 seg = 1ull << kn.seg_log2;
    const u64 nseg = (n + seg - 1) / seg;
    const u64 tiles = scan_tiles(n);
    const u64 M = kn.bucket_cap;                                   // elements a bucket may hold
    // the buckets' sort buffers also serve the rounds over the tied list, which take it in parts of half a buffer (the other half
    // holds the part's regrouped elements): short inputs (the forced runs of the tests) keep room for the whole list in one part
    const u64 whole = 2 * n < (1ull << 29) ? 2 * n : (1ull << 29);
    const u64 Mb = M > whole ? M : whole;
    const u64 mwords = (Mb + 63) / 64 + 1;
    u64 sort_max = Mb > seg ? Mb : seg;                            // largest sort or scan through tile_hist / scan_temp:
    if (sort_max < LYN_CAND_CAP) sort_max = LYN_CAND_CAP;          // a bucket, a segment, or the factor candidates
    // ---- arena layout -------------------------------------------------------------------------------------------------
    const size_t need = (direct ? 0 : align_up(n * 8, 256)) + align_up(seg * 8, 256) + align_up(seg, 256) + align_up((tiles + 1) * 8, 256) + scan_temp_bytes(n) +
                        2 * align_up(Mb * 8, 256) + 2 * align_up(Mb * 4, 256) + 4 * align_up(Mb, 256) + radix_tile_hist_bytes(sort_max) +
                        scan_temp_bytes(sort_max) + 3 * align_up(mwords * 8, 256) + 8 * align_up(LYN_CAND_CAP * 8, 256) +
                        align_up(nseg * WIDE_PREFIXES * 4, 256) + align_up((WIDE_MAX_PARTS + 2) * 8, 256) + (1 << 16);
    BWTS_TRY(arena_reserve(ctx, need));
    u64 *rank64 = direct ? nullptr : arena_array<u64>(ctx, n);
    u64 *segkeys = arena_array<u64>(ctx, seg);
    u8 *segprev = arena_array<u8>(ctx, seg);
    u64 *tile_min = arena_array<u64>(ctx, tiles + 1);
    void *pre_temp = arena_alloc(ctx, scan_temp_bytes(n));
    u64 *bk[2] = {arena_array<u64>(ctx, Mb), arena_array<u64>(ctx, Mb)};
    u32 *bv[2] = {arena_array<u32>(ctx, Mb), arena_array<u32>(ctx, Mb)};
    u8 *bs_src = arena_array<u8>(ctx, Mb), *bs_buf[2] = {arena_array<u8>(ctx, Mb), arena_array<u8>(ctx, Mb)}, *bs_fin = arena_array<u8>(ctx, Mb);
    u32 *tile_hist = (u32 *)arena_alloc(ctx, radix_tile_hist_bytes(sort_max));
    void *scan_temp = arena_alloc(ctx, scan_temp_bytes(sort_max));
    u64 *headw = arena_array<u64>(ctx, mwords), *keepw = arena_array<u64>(ctx, mwords), *prew = arena_array<u64>(ctx, mwords);
    u64 *cand[2] = {arena_array<u64>(ctx, LYN_CAND_CAP), arena_array<u64>(ctx, LYN_CAND_CAP)};
    u32 *cvals[2] = {arena_array<u32>(ctx, LYN_CAND_CAP), arena_array<u32>(ctx, LYN_CAND_CAP)};
    u64 *fstart = arena_array<u64>(ctx, LYN_CAND_CAP);
    u32 *d_hist = arena_array<u32>(ctx, nseg * WIDE_PREFIXES);
    u64 *d_cuts = arena_array<u64>(ctx, WIDE_MAX_PARTS + 2);
    u64 **d_tab = (u64 **)arena_array<u64>(ctx, WIDE_TB_MAX);
    if (!d_cuts || !d_tab) return BWTS_E_NOMEM;
    if ((!direct && !rank64) || !segkeys || !segprev || !tile_min || !pre_temp || !bk[1] || !bv[1] || !bs_src || !bs_buf[1] || !bs_fin || !tile_hist || !scan_temp ||
        !headw || !keepw || !prew || !cand[1] || !cvals[1] || !fstart || !d_hist)
        return BWTS_E_NOMEM;

    // ---- alphabet and key format ---------------------------------------------------------------------------------------
    BWTS_TRY(read_histogram(ctx, d_T, n));
    Alphabet al;
    BWTS_TRY(set_alphabet(ctx, false, n, &al));
    ctx->tm.key_symbols = (u32)al.msym;
    ctx->tm.key_bits = (u32)al.key_bits;
    const int pbits = al.key_bits < WIDE_PREFIX_BITS ? al.key_bits : WIDE_PREFIX_BITS;
    const int pshift = al.key_bits - pbits;
    const u64 *d_vtab = al.varlen ? ctx->d_small + SM_VTAB : nullptr;
    const u8 *d_codes = (const u8 *)(ctx->d_small + SM_CODES);
    auto seg_count = [&](u64 s) -> u64 { const u64 p0 = s * seg; return n - p0 < seg ? n - p0 : seg; };

    // ---- Lyndon factors (mk_bwts_sa.c:126-129), candidate search as on the main path, segment by segment -----------------
    for (u64 s = 0; s < nseg; s++)
        BWTS_TRY(launch_keybuild0_seg(ctx, d_T, n, al, segkeys, tile_min + s * (seg / KB_TILE), false, s * seg, seg_count(s)));
    u64 *cnt = ctx->d_small + SM_COUNTERS;
    if (direct && kn.direct < 0) {
        // is this an input with few ties at all?  2^20 of the last segment's keys (still in segkeys), sorted: text shows 10^5 equal
        // neighbours, i.i.d. data none.  Two or more: the rank array is needed, and the caller is told before a bucket is sorted.
        const u64 c = seg_count(nseg - 1);
        const u32 m = c >= (1u << 20) ? (1u << 20) : (u32)c;
        if (m >= 4096 && m <= Mb) {
            HIPC(hipMemsetAsync(cnt + 27, 0, sizeof(u64), ctx->stream));
            wide_sample_keys_kernel<<<dim3((m + 255) / 256), dim3(256), 0, ctx->stream>>>(segkeys, c, m, bk[0], bv[0]);
            HIPC(hipGetLastError());
            SortPlan spn;
            spn.keys[0] = bk[0]; spn.keys[1] = bk[1];
            spn.vals[0] = bv[0]; spn.vals[1] = bv[1];
            spn.tile_hist = tile_hist; spn.scan_temp = scan_temp;
            int sres = 0;
            BWTS_TRY(radix_sort_pairs(ctx, spn, m, al.key_bits, &sres));
            wide_sample_ties_kernel<<<dim3((m + 255) / 256), dim3(256), 0, ctx->stream>>>(spn.keys[sres], m, cnt + 27);
            HIPC(hipGetLastError());
            BWTS_TRY(read_small(ctx, SM_COUNTERS + 27, 1));
            if (ctx->h_small[SM_COUNTERS + 27] >= 2) { *again = true; return BWTS_OK; }
        }
    }
    HIPC(hipMemsetAsync(cnt + 4, 0, 4 * sizeof(u64), ctx->stream));
    {
        SpanGuard g(ctx, BWTS_K_LYNDON, n, 0);
        // exclusive prefix minima of the tile minima, in pre_temp (laid out like the partials of a scan over n elements)
        HIPC(hipMemcpyAsync(pre_temp, tile_min, tiles * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
        BWTS_TRY((device_scan_partials<u64, OpMin>(ctx, tiles, OpMin(), ~0ull, pre_temp)));
    }
    for (u64 s = 0; s < nseg; s++) {
        const u64 p0 = s * seg, c = seg_count(s), t0 = p0 / KB_TILE;
        BWTS_TRY(launch_keybuild0_seg(ctx, d_T, n, al, segkeys, nullptr, false, p0, c));
        SpanGuard g(ctx, BWTS_K_LYNDON, c, 8 * c);
        const KeyStore ks = key_store_of(segkeys, c, false, al.key_bits);
        KeyIn in{ks};
        CandOut out{ks, n, al.varlen ? 64 : al.msym, cand[0], LYN_CAND_CAP, ctx->d_small + CNT_CAND, p0};
        TileMayHoldCandidate filter{tile_min + t0};
        BWTS_TRY((device_scan_final<false, u64>(ctx, c, in, out, OpMin(), ~0ull, (u64 *)pre_temp + t0, filter)));
    }
    BWTS_TRY(read_small(ctx, CNT_CAND, 1));
    const u64 cnt_c = ctx->h_small[CNT_CAND];
    if (cnt_c == 0) return BWTS_E_INTERNAL;
    if (cnt_c > LYN_CAND_CAP) return BWTS_E_RANGE;           // (a^n, (ab)^n ...: the suffix-sort route to the factors has no wide form)
    u64 k = 0;
    {
        SortPlan cp;
        cp.keys[0] = cand[0]; cp.keys[1] = cand[1];
        cp.vals[0] = cvals[0]; cp.vals[1] = cvals[1];
        cp.tile_hist = tile_hist; cp.scan_temp = scan_temp;
        int res = 0;
        BWTS_TRY(radix_sort_pairs(ctx, cp, cnt_c, bitlen_u64(n) + 1, &res));
        LynState *d_st = (LynState *)(ctx->d_small + CNT_LYN_K);
        LynState *h_st = (LynState *)(ctx->h_small + CNT_LYN_K);
        unsigned long long *d_mis = (unsigned long long *)(ctx->d_small + CNT_LYN_K + 8);
        HIPC(hipMemsetAsync(d_st, 0, sizeof(LynState), ctx->stream));
        for (int iter = 0;; iter++) {
            {
                SpanGuard g(ctx, BWTS_K_LYNDON, cnt_c, 0);
                lyndon_resolve_kernel<u64><<<dim3(1), dim3(256), 0, ctx->stream>>>(d_T, n, cand[res], cnt_c, fstart, d_st, LYN_WORK_CAP);
                HIPC(hipGetLastError());
            }
            BWTS_TRY(read_small(ctx, CNT_LYN_K, 8));
            if (h_st->status == 0) break;
            if (h_st->status == 2 || iter > 4096) return BWTS_E_RANGE;
            u64 e = 0;
            HIPC(hipMemcpyAsync(&e, cand[res] + h_st->next, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
            HIPC(hipStreamSynchronize(ctx->stream));
            const u64 pp = e >> 1, qq = h_st->cur;
            const u64 len = n - pp;
            HIPC(hipMemcpyAsync(d_mis, &len, sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
            {
                SpanGuard g(ctx, BWTS_K_LYNDON, len, 2 * len);
                u64 blocks = (len + 255) / 256; if (blocks > 4096) blocks = 4096;
                suffix_mismatch_grid_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(d_T, n, pp, qq, d_mis);
                HIPC(hipGetLastError());
            }
            BWTS_TRY(read_small(ctx, CNT_LYN_K + 8, 1));
            const u64 at = ctx->h_small[CNT_LYN_K + 8];
            bool less;
            if (at >= len) less = true;
            else {
                u8 ab[2];
                HIPC(hipMemcpyAsync(&ab[0], d_T + pp + at, 1, hipMemcpyDeviceToHost, ctx->stream));
                HIPC(hipMemcpyAsync(&ab[1], d_T + qq + at, 1, hipMemcpyDeviceToHost, ctx->stream));
                HIPC(hipStreamSynchronize(ctx->stream));
                less = ab[0] < ab[1];
            }
            h_st->forced = 1; h_st->forced_less = less ? 1 : 0; h_st->status = 0;
            HIPC(hipMemcpyAsync(d_st, h_st, sizeof(LynState), hipMemcpyHostToDevice, ctx->stream));
            HIPC(hipStreamSynchronize(ctx->stream));
        }
        k = h_st->k;
        if (k == 0) return BWTS_E_INTERNAL;
    }
    ctx->tm.factors = k;
    const PrevSym64 prev{d_T, n, fstart, k};
    const bool carry_ok = [ctx] { const char *e = bwts_knob(ctx, "BWTS_WIDE_CARRY"); return !(e && atoi(e) == 0); }();
    // the output byte travels with the sort (see WideFilterOut): the passes cover whole bytes of the key, so the parked bits must lie above them
    const bool carry = carry_ok && 8 * ((al.key_bits + 7) / 8) <= WIDE_HI_SHIFT;

    // cyclic round-0 keys of one segment: keybuild + the wrap-around patch near the factor ends
    auto seg_keys = [&](u64 s) -> int {
        const u64 p0 = s * seg, c = seg_count(s);
        BWTS_TRY(launch_keybuild0_seg(ctx, d_T, n, al, segkeys, nullptr, false, p0, c, carry ? segprev : nullptr));
        const int span = al.varlen ? 64 : al.msym - 1;
        if (span > 0) {
            const u64 threads = k * (u64)span;
            cyclic_patch_wide_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream>>>(
                d_T, n, d_codes, al.bits, al.msym, d_vtab, al.key_bits, span, fstart, k, segkeys, p0, p0 + c);
            HIPC(hipGetLastError());
        }
        if (carry) {
            wide_head_prev_kernel<<<dim3((unsigned)((k + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_T, n, fstart, k, segprev, p0, p0 + c);
            HIPC(hipGetLastError());
        }
        return BWTS_OK;
    };

    // ---- bucket sizes: histogram of the key prefixes, per segment --------------------------------------------------------
    HIPC(hipMemsetAsync(d_hist, 0, nseg * WIDE_PREFIXES * sizeof(u32), ctx->stream));
    for (u64 s = 0; s < nseg; s++) {
        BWTS_TRY(seg_keys(s));
        const u64 c = seg_count(s);
        u64 blocks = (c + 255) / 256; if (blocks > 2048) blocks = 2048;
        SpanGuard g(ctx, BWTS_K_HISTOGRAM, c, 8 * c);
        wide_prefix_hist_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(segkeys, c, pshift, d_hist + s * WIDE_PREFIXES);
        HIPC(hipGetLastError());
    }
    std::vector<u32> h_hist((size_t)nseg * WIDE_PREFIXES);
    HIPC(hipMemcpyAsync(h_hist.data(), d_hist, h_hist.size() * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    const u32 nprefix = 1u << pbits;
    std::vector<u64> ptotal(nprefix, 0);
    for (u64 s = 0; s < nseg; s++) for (u32 q = 0; q < nprefix; q++) ptotal[q] += h_hist[(size_t)s * WIDE_PREFIXES + q];
    // greedy: consecutive prefixes while the bucket stays within M
    std::vector<u32> cut;                 // bucket b = prefixes [cut[b], cut[b + 1])
    {
        u64 run = 0;
        cut.push_back(0);
        for (u32 q = 0; q < nprefix; q++) {
            if (ptotal[q] > M) return BWTS_E_RANGE;        // one key prefix holds more positions than a bucket may: no finer cut in this form
            if (run + ptotal[q] > M) { cut.push_back(q); run = 0; }
            run += ptotal[q];
        }
        cut.push_back(nprefix);
    }

    // ---- tied list: blocks taken as it grows ---------------------------------------------------------------------------------
    int tlg = kn.tblock_log2;
    if (!tlg) { tlg = bitlen_u64(n - 1); if (tlg < 8) tlg = 8; if (tlg > 28) tlg = 28; }
    BWTS_TRY(wide_tied_ensure(ctx, 1, tlg, d_tab, true));
    TiedList tl{d_tab, ctx->tied_blk_lg};
    u64 *d_tied = cnt + 24, *d_over = cnt + 25;
    HIPC(hipMemsetAsync(cnt + 24, 0, 2 * sizeof(u64), ctx->stream));
    u64 tied_total = 0;

    // ---- bucket by bucket ----------------------------------------------------------------------------------------------------
    u64 base = 0;
    for (size_t b = 0; b + 1 < cut.size(); b++) {
        const u32 plo = cut[b], phi = cut[b + 1];
        u64 m = 0;
        for (u32 q = plo; q < phi; q++) m += ptotal[q];
        if (m == 0) continue;
        u64 off = 0;
        for (u64 s = 0; s < nseg; s++) {
            u64 share = 0;
            for (u32 q = plo; q < phi; q++) share += h_hist[(size_t)s * WIDE_PREFIXES + q];
            if (share == 0) continue;
            BWTS_TRY(seg_keys(s));
            const u64 c = seg_count(s);
            SpanGuard g(ctx, BWTS_K_RERANK, c, 9 * c + 13 * share);
            WideFilterIn fin{segkeys, pshift, plo, phi};
            WideFilterOut fout{segkeys, pshift, plo, phi, s * seg, bk[0] + off, bv[0] + off, bs_src + off, carry ? segprev : nullptr};
            BWTS_TRY((device_scan<false, u32>(ctx, c, fin, fout, OpAdd(), 0u, scan_temp)));
            off += share;
        }
        if (off != m) return BWTS_E_INTERNAL;
        SortPlan sp;
        sp.keys[0] = bk[0]; sp.keys[1] = bk[1];
        sp.vals[0] = bv[0]; sp.vals[1] = bv[1];
        sp.tile_hist = tile_hist; sp.scan_temp = scan_temp;
        sp.sym_src = bs_src; sp.sym_buf[0] = bs_buf[0]; sp.sym_buf[1] = bs_buf[1]; sp.sym_final = bs_fin;
        int res = 0;
        BWTS_TRY(radix_sort_pairs(ctx, sp, m, al.key_bits, &res));
        const u64 words = (m + 63) / 64;
        {
            SpanGuard g(ctx, BWTS_K_RERANK, m, 8 * m);
            u64 waves = (words + GF_WORDS - 1) / GF_WORDS;
            unsigned blocks = (unsigned)((waves + 3) / 4 < 16384 ? (waves + 3) / 4 : 16384);
            group_flags_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(sp.keys[res], m, headw, keepw, carry ? (1ull << WIDE_HI_SHIFT) - 1ull : ~0ull);
            WordIn win{headw, keepw};
            ScanStoreArr<u64> wout{prew};
            BWTS_TRY((device_scan<false, u64>(ctx, words, win, wout, OpHeadCount(), (u64)0, scan_temp)));
        }
        // the bucket's tied elements are counted before they are written: the list takes another block when they need one
        wide_add_tied_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(keepw, prew, words, d_tied);
        HIPC(hipGetLastError());
        BWTS_TRY(read_small(ctx, SM_COUNTERS + 24, 1));
        const u64 tied_before = tied_total;
        tied_total = ctx->h_small[SM_COUNTERS + 24];
        if (tied_total < tied_before || tied_total - tied_before > m) return BWTS_E_INTERNAL;
        if (direct && tied_total > WIDE_DIRECT_MAX) { *again = true; return BWTS_OK; }          // many ties: this input needs the ranks
        BWTS_TRY(wide_tied_ensure(ctx, tied_total ? tied_total : 1, tlg, d_tab, false));
        {
            SpanGuard g(ctx, BWTS_K_EMIT, m, 15 * m);
            u64 blocks = (m + 255) / 256; if (blocks > 16384) blocks = 16384;
            if (carry)
                wide_bucket_finish_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(sp.keys[res], sp.vals[res], bs_fin, m, base, headw, keepw, prew,
                                                                                                      rank64, prev, d_out, tied_before, tied_total, tl, d_over, n);
            else
                wide_bucket_finish_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(sp.keys[res], sp.vals[res], bs_fin, m, base, headw, keepw, prew,
                                                                                                       rank64, prev, d_out, tied_before, tied_total, tl, d_over, n);
            HIPC(hipGetLastError());
        }
        base += m;
    }
    if (base != n) return BWTS_E_INTERNAL;
    BWTS_TRY(read_small(ctx, SM_COUNTERS + 24, 2));
    u64 a = ctx->h_small[SM_COUNTERS + 24];
    if (ctx->h_small[SM_COUNTERS + 25] == 2) return BWTS_E_INTERNAL;              // a sorted element carried a position outside the text
    if (ctx->h_small[SM_COUNTERS + 25] || a != tied_total) return BWTS_E_INTERNAL; // (the list had room for every count read above)
    ctx->tm.active_after_round0 = a;
    ctx->tm.round_active[0] = a;
    if (direct) {
        if (a) {
            if (a > WIDE_DIRECT_MAX) { *again = true; return BWTS_OK; }
            SpanGuard g(ctx, BWTS_K_RERANK, a, 40 * a);
            u32 *gstart = bv[0];                         // (a <= WIDE_DIRECT_MAX <= the buffers' size)
            if (a > Mb) return BWTS_E_INTERNAL;
            HIPC(hipMemsetAsync(cnt, 0, 4 * sizeof(u64), ctx->stream));
            WideGroupIn gin{tl};
            WideGroupOut gout{tl, gstart, a, cnt + 3};
            BWTS_TRY((device_scan<false, u32>(ctx, a, gin, gout, OpAdd(), 0u, scan_temp)));
            BWTS_TRY(read_small(ctx, SM_COUNTERS, 4));
            const u64 ngroups = ctx->h_small[SM_COUNTERS + 3];
            if (ngroups == 0 || ngroups > a) return BWTS_E_INTERNAL;
            wide_direct_groups_kernel<<<dim3((unsigned)((ngroups + 63) / 64)), dim3(64), 0, ctx->stream>>>(tl, gstart, ngroups, a, d_T, n, fstart, k, (u64)al.hstep, prev,
                                                                                                          d_out, cnt + 1);
            HIPC(hipGetLastError());
            BWTS_TRY(read_small(ctx, SM_COUNTERS, 4));
            if (ctx->h_small[SM_COUNTERS + 1]) { *again = true; return BWTS_OK; }               // a large group, or rotations equal for thousands of symbols
        }
        ctx->tm.rounds = a ? 2 : 1;
        if (a && 1 < BWTS_MAX_ROUND_STATS) ctx->tm.round_active[1] = 0;
        return BWTS_OK;
    }

    // ---- rounds over the tied list, part by part ----------------------------------------------------------------------------
    u32 rounds = 1;
    const int rb = bitlen_u64(n - 1);
    u64 P = kn.part ? kn.part : Mb / 2;
    if (P > Mb / 2) P = Mb / 2;
    while (P > 64 && bitlen_u64(P / 2) + rb > 64) P >>= 1;       // (group ordinal, rank) must fit a 64-bit sort key
    u64 *npos = bk[0] + (Mb - Mb / 2), *nhead = bk[1] + (Mb - Mb / 2);       // the buffers' second halves
    u8 *tstate = bs_src;
    u64 *d_wp = cnt + 26, *d_cnt = cnt + 0, *d_split = cnt + 1, *d_groups = cnt + 3;
    std::vector<u64> h_cuts(WIDE_MAX_PARTS + 2);
    auto cut_parts = [&](u64 count, u64 *nparts) -> int {
        wide_cuts_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(tl, count, P, d_cuts, WIDE_MAX_PARTS);
        HIPC(hipGetLastError());
        HIPC(hipMemcpyAsync(h_cuts.data(), d_cuts, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));
        if (h_cuts[0] == ~0ull) return BWTS_E_RANGE;               // a group of more elements than a part holds
        *nparts = h_cuts[0];
        HIPC(hipMemcpyAsync(h_cuts.data() + 1, d_cuts + 1, (size_t)*nparts * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));
        return BWTS_OK;
    };
    for (u64 h = (u64)al.hstep; a > 0; h <<= 1) {
        rounds++;
        u64 nparts = 0;
        BWTS_TRY(cut_parts(a, &nparts));
        HIPC(hipMemsetAsync(cnt, 0, 4 * sizeof(u64), ctx->stream));
        HIPC(hipMemsetAsync(d_wp, 0, sizeof(u64), ctx->stream));
        u64 lo = 0;
        for (u64 part = 0; part < nparts; part++) {
            const u64 e = h_cuts[1 + part];
            if (e <= lo || e > a || e - lo > P) return BWTS_E_INTERNAL;
            const u64 m = e - lo;
            {
                SpanGuard g(ctx, BWTS_K_KEYBUILD, m, 36 * m);
                WideOrdIn oin{tl, lo};
                WideOrdOut oout{tl, lo, rank64, n, h, fstart, k, rb, bk[0], bv[0], m, d_groups};
                BWTS_TRY((device_scan<false, u32>(ctx, m, oin, oout, OpAdd(), 0u, scan_temp)));
            }
            BWTS_TRY(read_small(ctx, SM_COUNTERS + 3, 1));
            const u64 groups = ctx->h_small[SM_COUNTERS + 3];
            if (bitlen_u64(groups) + rb > 64) return BWTS_E_RANGE;
            SortPlan tp;
            tp.keys[0] = bk[0]; tp.keys[1] = bk[1];
            tp.vals[0] = bv[0]; tp.vals[1] = bv[1];
            tp.tile_hist = tile_hist; tp.scan_temp = scan_temp;
            int tres = 0;
            BWTS_TRY(radix_sort_pairs(ctx, tp, m, bitlen_u64(groups) + rb, &tres));
            {
                SpanGuard g(ctx, BWTS_K_RERANK, m, 48 * m);
                DgRegroupIn rin{bk[tres], m, rb};
                WideRegroupOut rout{bk[tres], bv[tres], m, tl, lo, npos, nhead, tstate, prev, d_out, d_split};
                BWTS_TRY((device_scan<true, u64>(ctx, m, rin, rout, OpMax2(), (u64)0, scan_temp)));
                WideKeepIn kin{tstate};
                WideKeepOut kout{tstate, npos, nhead, rank64, tl, d_wp, m, d_cnt};
                BWTS_TRY((device_scan<false, u32>(ctx, m, kin, kout, OpAdd(), 0u, scan_temp)));
                wide_advance_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(d_wp, d_cnt);
                HIPC(hipGetLastError());
            }
            lo = e;
        }
        if (lo != a) return BWTS_E_INTERNAL;
        BWTS_TRY(read_small(ctx, SM_COUNTERS, 4));
        BWTS_TRY(read_small(ctx, SM_COUNTERS + 26, 1));
        const u64 a_new = ctx->h_small[SM_COUNTERS + 26], splits = ctx->h_small[SM_COUNTERS + 1];
        if (a_new > a) return BWTS_E_INTERNAL;
        a = a_new;
        if (rounds - 1 < BWTS_MAX_ROUND_STATS) ctx->tm.round_active[rounds - 1] = a;
        if (a == 0 || splits == 0) break;                   // no group split: equal infinite words
        if (rounds > 80) return BWTS_E_INTERNAL;
    }
    if (a) {
        u64 nparts = 0;
        BWTS_TRY(cut_parts(a, &nparts));
        u64 lo = 0;
        for (u64 part = 0; part < nparts; part++) {
            const u64 e = h_cuts[1 + part], m = e - lo;
            SpanGuard g(ctx, BWTS_K_EMIT, m, 20 * m);
            WideRestIn rin{tl, lo};
            WideRestOut rout{tl, lo, prev, d_out};
            BWTS_TRY((device_scan<true, u32>(ctx, m, rin, rout, OpMax(), 0u, scan_temp)));
            lo = e;
        }
    }
    ctx->tm.rounds = rounds;
    return BWTS_OK;
}

static int forward_wide_impl(bwts_ctx *ctx, const u8 *d_T, u64 n, u8 *d_out)
{
    const WideKnobs kn = wide_knobs(ctx);
    bool again = false;
    if (kn.direct != 0) {
        // first without the rank array: an input with few ties never needs it (8 n bytes, n random writes); one with many says so
        // after a look at a sample of its keys, or at the latest when its tied list passes WIDE_DIRECT_MAX
        const int rc = forward_wide_run(ctx, d_T, n, d_out, true, &again);
        if (rc != BWTS_OK || !again) return rc;
        if (kn.direct == 1) return BWTS_E_RANGE;          // (forced: the tests want to know that this form did it)
    }
    return forward_wide_run(ctx, d_T, n, d_out, false, &again);
}
