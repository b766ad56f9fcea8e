// forward.hip -- BWTS forward transform on the GPU.
//
// Replaces divsufsort() + make_bwts_sa() + move_lyndonword_head()
// (/root/reference/mk_bwts_sa.c:48, :114-195, :74-112).  Instead of suffix-sorting and then
// patching SA/ISA sequentially, the engine
//   1. finds the Lyndon factors (prefix minima of suffix ranks, mk_bwts_sa.c:126-129),
//   2. sorts all positions directly by their infinite cyclic word rot(p)^omega with prefix
//      doubling on a CYCLIC successor (no fix-up needed: the result IS the fixed-up order),
//   3. emits bwts[r] = T[cprev(sa[r])] (mk_bwts_sa.c:172-188) as a gather.
// All index work is u32 with u64 loop bounds; bytes are unsigned.
#include "internal.h"
#include "device_utils.h"
#include "scan_templ.h"

#include <stdlib.h>
#include <string.h>

// ------------------------------------------------------------------------------------
// small-word layout in ctx->d_small / h_small (u64 words)
// ------------------------------------------------------------------------------------
#define SM_HIST      0      // 256 words: byte histogram
#define SM_CODES     256    // 32 words = 256 bytes: byte -> symbol code
#define SM_COUNTERS  320    // scratch counters
#define   CNT_ACTIVE   (SM_COUNTERS + 0)
#define   CNT_SPLITS   (SM_COUNTERS + 1)
#define   CNT_TOTAL    (SM_COUNTERS + 2)

struct Alphabet {
    int sigma;      // distinct byte values present
    int bits;       // bits per symbol code
    int msym;       // symbols packed into a round-0 key
    int key_bits;   // bits * msym
};

static int bitlen_u64(u64 x) { int b = 0; while (x) { b++; x >>= 1; } return b; }

// ------------------------------------------------------------------------------------
// byte histogram (also used by the inverse: unbwts.c:34-36)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void byte_hist_kernel(const u8 *__restrict__ T, u64 n, u64 *__restrict__ hist)
{
    __shared__ u32 bins[4][256];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < 1024; i += 256) ((u32 *)bins)[i] = 0;
    __syncthreads();
    // 16 bytes per thread per step when aligned; tail and head handled bytewise
    const u64 nvec = n / 16;
    const uint4 *T16 = (const uint4 *)T;   // hipMalloc'd / arena pointers are 256-B aligned
    const bool aligned = ((uintptr_t)T & 15) == 0;
    if (aligned) {
        for (u64 v = (u64)blockIdx.x * 256 + tid; v < nvec; v += (u64)gridDim.x * 256) {
            const uint4 q = T16[v];
            const u32 ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int b = 0; b < 4; b++) atomicAdd(&bins[w][(ws[a] >> (8 * b)) & 255u], 1u);
            }
        }
        for (u64 i = nvec * 16 + (u64)blockIdx.x * 256 + tid; i < n; i += (u64)gridDim.x * 256) atomicAdd(&bins[w][T[i]], 1u);
    } else {
        for (u64 i = (u64)blockIdx.x * 256 + tid; i < n; i += (u64)gridDim.x * 256) atomicAdd(&bins[w][T[i]], 1u);
    }
    __syncthreads();
    const u32 s = bins[0][tid] + bins[1][tid] + bins[2][tid] + bins[3][tid];
    if (s) atomicAdd((unsigned long long *)&hist[tid], (unsigned long long)s);
}

int byte_histogram_device(bwts_ctx *ctx, const u8 *d_T, u64 n, u64 *d_hist256)
{
    HIPC(hipMemsetAsync(d_hist256, 0, 256 * sizeof(u64), ctx->stream));
    u64 blocks = (n + 256 * 64 - 1) / (256 * 64);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    SpanGuard g(ctx, BWTS_K_HISTOGRAM, n, n);
    byte_hist_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(d_T, n, d_hist256);
    HIPC(hipGetLastError());
    return BWTS_OK;
}

// builds the byte->code table: cyclic sort uses codes 0..sigma-1, the non-cyclic (suffix)
// sort reserves code 0 for "past the end" and uses 1..sigma
static int make_alphabet(bwts_ctx *ctx, const u8 *d_T, u64 n, bool reserve_pad, Alphabet *al)
{
    BWTS_TRY(byte_histogram_device(ctx, d_T, n, ctx->d_small + SM_HIST));
    BWTS_TRY(read_small(ctx, SM_HIST, 256));
    u8 codes[256];
    int sigma = 0;
    for (int c = 0; c < 256; c++) {
        codes[c] = 0;
        if (ctx->h_small[SM_HIST + c]) {
            codes[c] = (u8)(sigma + (reserve_pad ? 1 : 0));   // sigma == 256 with pad handled below
            sigma++;
        }
    }
    int ncodes = sigma + (reserve_pad ? 1 : 0);
    int bits = bitlen_u64((u64)(ncodes > 1 ? ncodes - 1 : 1));
    if (bits > 8) {
        // 256 symbols + pad: 9-bit codes; the u8 table cannot hold code 256, so the
        // kernels add the +1 themselves (codes table then holds 0..255)
        for (int c = 0, s = 0; c < 256; c++) if (ctx->h_small[SM_HIST + c]) codes[c] = (u8)(s++);
    }
    al->sigma = sigma;
    al->bits = bits;
    al->msym = 64 / bits;
    const char *env = getenv("BWTS_KEY_SYMBOLS");
    if (env) { int v = atoi(env); if (v >= 1 && v < al->msym) al->msym = v; }
    al->key_bits = al->bits * al->msym;
    memcpy(ctx->h_small + SM_CODES, codes, 256);
    HIPC(hipMemcpyAsync(ctx->d_small + SM_CODES, ctx->h_small + SM_CODES, 256, hipMemcpyHostToDevice, ctx->stream));
    return BWTS_OK;
}

// ------------------------------------------------------------------------------------
// factor lookup: fstart[0..k) sorted factor starts, factor f = [fstart[f], f+1<k ? fstart[f+1] : n)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ u64 factor_of(const u32 *__restrict__ fstart, u64 k, u64 p)
{
    u64 lo = 0, hi = k - 1;
    while (lo < hi) {
        const u64 mid = (lo + hi + 1) >> 1;
        if ((u64)fstart[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ u64 factor_end(const u32 *__restrict__ fstart, u64 k, u64 n, u64 f)
{
    return f + 1 < k ? (u64)fstart[f + 1] : n;
}

// ------------------------------------------------------------------------------------
// round 0: key[p] = first msym symbols of the (cyclic | padded) word at p, value = p
// ------------------------------------------------------------------------------------
#define KB_THREADS 256
#define KB_ITEMS   8
#define KB_TILE    (KB_THREADS * KB_ITEMS)
#define KB_HALO    64

template <bool CYCLIC>
__global__ __launch_bounds__(KB_THREADS) void keybuild0_kernel(const u8 *__restrict__ T, u64 n, const u8 *__restrict__ codes_g,
                                                               int bits, int msym, int pad_add,
                                                               const u32 *__restrict__ fstart, u64 k,
                                                               u64 *__restrict__ keys, u32 *__restrict__ vals)
{
    __shared__ u16 sc[KB_TILE + KB_HALO];
    __shared__ u8 codes[256];
    __shared__ int fast_flag;

    const int tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * KB_TILE;
    const u64 end = base + KB_TILE < n ? base + KB_TILE : n;
    codes[tid] = codes_g[tid];
    if (tid == 0) {
        const u64 last_read = end + (u64)msym - 2;   // index of the last byte any key of this tile reads
        if (CYCLIC) {
            const u64 f = factor_of(fstart, k, base);
            fast_flag = last_read < factor_end(fstart, k, n, f);
        } else {
            fast_flag = last_read < n;
        }
    }
    __syncthreads();
    const int key_bits = bits * msym;
    const u64 mask = key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1ull);

    if (fast_flag) {
        const u32 span = (u32)(end - base) + (u32)msym - 1;
        for (u32 i = tid; i < span; i += KB_THREADS) sc[i] = (u16)((u32)codes[T[base + i]] + (u32)pad_add);
        __syncthreads();
        const u32 o = (u32)tid * KB_ITEMS;
        if (base + o < end) {
            u64 key = 0;
            for (int j = 0; j < msym; j++) key = (key << bits) | sc[o + j];
#pragma unroll
            for (int e = 0; e < KB_ITEMS; e++) {
                const u64 p = base + o + e;
                if (p < end) {
                    keys[p] = key;
                    vals[p] = (u32)p;
                    key = ((key << bits) | sc[o + e + msym]) & mask;   // sc read stays inside the halo
                }
            }
        }
    } else {
        for (int e = 0; e < KB_ITEMS; e++) {
            const u64 p = base + (u64)tid * KB_ITEMS + e;
            if (p >= end) break;
            u64 key = 0;
            if (CYCLIC) {
                const u64 f = factor_of(fstart, k, p);
                const u64 s = fstart[f], fe = factor_end(fstart, k, n, f);
                u64 q = p;
                for (int j = 0; j < msym; j++) {
                    key = (key << bits) | (u64)codes[T[q]];
                    if (++q == fe) q = s;
                }
            } else {
                for (int j = 0; j < msym; j++) {
                    const u64 q = p + j;
                    key = (key << bits) | (q < n ? (u64)codes[T[q]] + (u64)pad_add : 0ull);
                }
            }
            keys[p] = key;
            vals[p] = (u32)p;
        }
    }
}

// ------------------------------------------------------------------------------------
// round with step h: key = (group head, rank of the h-th (cyclic) successor)
// ------------------------------------------------------------------------------------
template <bool CYCLIC>
__global__ __launch_bounds__(256) void keybuild_h_kernel(const u32 *__restrict__ a_idx, const u32 *__restrict__ a_head, u64 a,
                                                         const u32 *__restrict__ rank, u64 n, u64 h, int rb,
                                                         const u32 *__restrict__ fstart, u64 k, u64 *__restrict__ keys)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= a) return;
    const u64 p = a_idx[i];
    u64 r2;
    if (CYCLIC) {
        const u64 f = factor_of(fstart, k, p);
        const u64 s = fstart[f], L = factor_end(fstart, k, n, f) - s;
        const u64 q = s + ((p - s) + h % L) % L;
        r2 = rank[q];
    } else {
        const u64 q = p + h;
        r2 = q < n ? (u64)rank[q] + 1ull : 0ull;
    }
    keys[i] = ((u64)a_head[i] << rb) | r2;
}

// ------------------------------------------------------------------------------------
// re-rank: group boundaries of the sorted keys -> new group heads, ranks, suffix array slots
// ------------------------------------------------------------------------------------
// slot(i): SA slot of the i-th sorted element (identity in round 0)
struct HeadIn {
    const u64 *K; const u32 *S;
    __device__ __forceinline__ u32 operator()(u64 i) const
    {
        const bool flag = i == 0 || K[i] != K[i - 1];
        return flag ? (S ? S[i] : (u32)i) : 0u;
    }
};

struct HeadOut {
    const u64 *K; const u32 *S; const u32 *V; u64 a; int rb;   // rb < 0: round 0 (no old groups)
    u32 *H; u32 *rank; u32 *SA;
    u64 *cnt_active, *cnt_splits;
    __device__ __forceinline__ void operator()(u64 i, u32 head) const
    {
        const u64 ki = K[i];
        const bool f0 = i == 0 || ki != K[i - 1];
        const bool f1 = i + 1 == a || K[i + 1] != ki;
        const u32 v = V[i];
        H[i] = head;
        rank[v] = head;
        if (S) SA[S[i]] = v;
        const bool keep = !(f0 && f1);
        const bool split = rb >= 0 && i > 0 && f0 && (ki >> rb) == (K[i - 1] >> rb);
        const u64 mk = __ballot(keep), ms = __ballot(split);
        if (lane_id() == 0) {
            if (mk) atomicAdd((unsigned long long *)cnt_active, (unsigned long long)__popcll(mk));
            if (ms) atomicAdd((unsigned long long *)cnt_splits, (unsigned long long)__popcll(ms));
        }
    }
};

struct KeepIn {
    const u64 *K; u64 a;
    __device__ __forceinline__ u32 operator()(u64 i) const
    {
        const u64 ki = K[i];
        const bool f0 = i == 0 || ki != K[i - 1];
        const bool f1 = i + 1 == a || K[i + 1] != ki;
        return (f0 && f1) ? 0u : 1u;
    }
};

struct KeepOut {
    const u64 *K; const u32 *S; const u32 *V; const u32 *H; u64 a;
    u32 *n_idx, *n_slot, *n_head;
    __device__ __forceinline__ void operator()(u64 i, u32 dst) const
    {
        const u64 ki = K[i];
        const bool f0 = i == 0 || ki != K[i - 1];
        const bool f1 = i + 1 == a || K[i + 1] != ki;
        if (!(f0 && f1)) {
            n_idx[dst] = V[i];
            n_slot[dst] = S ? S[i] : (u32)i;
            n_head[dst] = H[i];
        }
    }
};

// ------------------------------------------------------------------------------------
// the doubling sort
// ------------------------------------------------------------------------------------
struct SortSpace {
    u64 *keys[2];     // n each
    u32 *vals[2];     // n each
    u32 *rank;        // n
    u32 *tile_hist;
    void *scan_temp;
};

static size_t sort_space_bytes(u64 n)
{
    return 2 * align_up(n * 8, 256) + 3 * align_up(n * 4, 256) + radix_tile_hist_bytes(n) + scan_temp_bytes(n) + 4096;
}

static int sort_space_alloc(bwts_ctx *ctx, u64 n, SortSpace *sp)
{
    sp->keys[0] = arena_array<u64>(ctx, n);
    sp->keys[1] = arena_array<u64>(ctx, n);
    sp->vals[0] = arena_array<u32>(ctx, n);
    sp->vals[1] = arena_array<u32>(ctx, n);
    sp->rank = arena_array<u32>(ctx, n);
    sp->tile_hist = (u32 *)arena_alloc(ctx, radix_tile_hist_bytes(n));
    sp->scan_temp = arena_alloc(ctx, scan_temp_bytes(n));
    if (!sp->keys[0] || !sp->keys[1] || !sp->vals[0] || !sp->vals[1] || !sp->rank || !sp->tile_hist || !sp->scan_temp)
        return BWTS_E_NOMEM;
    return BWTS_OK;
}

// aux buffers for the active list; allocated once `a` is known
struct ActiveSpace {
    u32 *idx_alt;      // second value buffer of the sort
    u32 *slot[2];
    u32 *head[2];
    u32 *H;
};


template <bool CYCLIC>
static int doubling_sort(bwts_ctx *ctx, const u8 *d_T, u64 n, const Alphabet &al, const u32 *d_fstart, u64 k,
                         SortSpace &sp, u32 **sa_out, u32 *rounds_out, u64 *active0_out)
{
    const u8 *d_codes = (const u8 *)(ctx->d_small + SM_CODES);
    const int pad_add = (!CYCLIC && al.bits > 8) ? 1 : 0;
    u64 *cnt = ctx->d_small + SM_COUNTERS;

    // ---- round 0 ------------------------------------------------------------------
    {
        SpanGuard g(ctx, BWTS_K_KEYBUILD, n, n + 12 * n);
        const u64 blocks = (n + KB_TILE - 1) / KB_TILE;
        keybuild0_kernel<CYCLIC><<<dim3((unsigned)blocks), dim3(KB_THREADS), 0, ctx->stream>>>(
            d_T, n, d_codes, al.bits, al.msym, pad_add, d_fstart, k, sp.keys[0], sp.vals[0]);
        HIPC(hipGetLastError());
    }
    SortPlan plan;
    plan.keys[0] = sp.keys[0]; plan.keys[1] = sp.keys[1];
    plan.vals[0] = sp.vals[0]; plan.vals[1] = sp.vals[1];
    plan.tile_hist = sp.tile_hist; plan.scan_temp = sp.scan_temp;
    int res = 0;
    BWTS_TRY(radix_sort_pairs(ctx, plan, n, al.key_bits, &res));
    u64 *K = sp.keys[res];
    u32 *SA = sp.vals[res];
    u32 *H0 = sp.vals[res ^ 1];

    HIPC(hipMemsetAsync(cnt, 0, 4 * sizeof(u64), ctx->stream));
    {
        SpanGuard g(ctx, BWTS_K_RERANK, n, 16 * n);
        HeadIn in{K, nullptr};
        HeadOut out{K, nullptr, SA, n, -1, H0, sp.rank, SA, cnt + 0, cnt + 1};
        BWTS_TRY((device_scan<true, u32>(ctx, n, in, out, OpMax(), 0u, sp.scan_temp)));
    }
    BWTS_TRY(read_small(ctx, SM_COUNTERS, 4));
    u64 a = ctx->h_small[CNT_ACTIVE];
    *active0_out = a;
    u32 rounds = 1;
    if (a == 0) { *sa_out = SA; *rounds_out = rounds; return BWTS_OK; }

    // ---- active list ------------------------------------------------------------------
    ActiveSpace as;
    {
        char *base = nullptr;
        const size_t each = align_up((size_t)a * 4, 256);
        BWTS_TRY(aux_reserve(ctx, 6 * each, &base));
        as.idx_alt = (u32 *)(base + 0 * each);
        as.slot[0] = (u32 *)(base + 1 * each);
        as.slot[1] = (u32 *)(base + 2 * each);
        as.head[0] = (u32 *)(base + 3 * each);
        as.head[1] = (u32 *)(base + 4 * each);
        as.H       = (u32 *)(base + 5 * each);
    }
    // the round-0 H buffer doubles as the first value buffer of the active sort once compacted;
    // compaction reads H0 while writing idx_alt, then the roles are fixed below
    u32 *idx_buf[2] = {as.idx_alt, H0};
    int ic = 0, sc = 0;
    {
        SpanGuard g(ctx, BWTS_K_RERANK, n, 12 * n);
        KeepIn in{K, n};
        KeepOut out{K, nullptr, SA, H0, n, idx_buf[0], as.slot[0], as.head[0]};
        BWTS_TRY((device_scan<false, u32>(ctx, n, in, out, OpAdd(), 0u, sp.scan_temp)));
    }

    const int rb = CYCLIC ? bitlen_u64(n - 1) : bitlen_u64(n);
    if (2 * rb > 64) return BWTS_E_RANGE;
    const int round_key_bits = 2 * rb > 1 ? 2 * rb : 1;

    for (u64 h = (u64)al.msym;; h <<= 1) {
        rounds++;
        {
            SpanGuard g(ctx, BWTS_K_KEYBUILD, a, 20 * a);
            keybuild_h_kernel<CYCLIC><<<dim3((unsigned)((a + 255) / 256)), dim3(256), 0, ctx->stream>>>(
                idx_buf[ic], as.head[sc], a, sp.rank, n, h, rb, d_fstart, k, sp.keys[0]);
            HIPC(hipGetLastError());
        }
        SortPlan ap;
        ap.keys[0] = sp.keys[0]; ap.keys[1] = sp.keys[1];
        ap.vals[0] = idx_buf[ic]; ap.vals[1] = idx_buf[ic ^ 1];
        ap.tile_hist = sp.tile_hist; ap.scan_temp = sp.scan_temp;
        int r2 = 0;
        BWTS_TRY(radix_sort_pairs(ctx, ap, a, round_key_bits, &r2));
        u64 *AK = sp.keys[r2];
        u32 *AV = r2 ? idx_buf[ic ^ 1] : idx_buf[ic];
        u32 *AVfree = r2 ? idx_buf[ic] : idx_buf[ic ^ 1];

        HIPC(hipMemsetAsync(cnt, 0, 4 * sizeof(u64), ctx->stream));
        {
            SpanGuard g(ctx, BWTS_K_RERANK, a, 24 * a);
            HeadIn in{AK, as.slot[sc]};
            HeadOut out{AK, as.slot[sc], AV, a, rb, as.H, sp.rank, SA, cnt + 0, cnt + 1};
            BWTS_TRY((device_scan<true, u32>(ctx, a, in, out, OpMax(), 0u, sp.scan_temp)));
        }
        BWTS_TRY(read_small(ctx, SM_COUNTERS, 4));
        const u64 a_new = ctx->h_small[CNT_ACTIVE];
        const u64 splits = ctx->h_small[CNT_SPLITS];
        if (a_new == 0) break;
        if (CYCLIC && splits == 0) break;            // partition stable under doubling: equal infinite words
        if (!CYCLIC && h >= n) return BWTS_E_INTERNAL; // suffixes are distinct; cannot happen
        if (rounds > 80) return BWTS_E_INTERNAL;
        {
            SpanGuard g(ctx, BWTS_K_RERANK, a, 24 * a);
            KeepIn in{AK, a};
            KeepOut out{AK, as.slot[sc], AV, as.H, a, AVfree, as.slot[sc ^ 1], as.head[sc ^ 1]};
            BWTS_TRY((device_scan<false, u32>(ctx, a, in, out, OpAdd(), 0u, sp.scan_temp)));
        }
        // the compacted list now lives in AVfree
        ic = (AVfree == idx_buf[0]) ? 0 : 1;
        sc ^= 1;
        a = a_new;
    }
    *sa_out = SA;
    *rounds_out = rounds;
    return BWTS_OK;
}

// ------------------------------------------------------------------------------------
// Lyndon factors = strict prefix minima of the suffix ranks (mk_bwts_sa.c:126-129)
// ------------------------------------------------------------------------------------
struct RankIn { const u32 *r; __device__ __forceinline__ u32 operator()(u64 i) const { return r[i]; } };
struct MinFlagOut {
    const u32 *r; u8 *flag;
    __device__ __forceinline__ void operator()(u64 i, u32 min_before) const { flag[i] = (i == 0 || r[i] < min_before) ? 1 : 0; }
};
struct FlagIn { const u8 *flag; __device__ __forceinline__ u32 operator()(u64 i) const { return flag[i]; } };
struct StartOut {
    const u8 *flag; u32 *starts; u64 n; u64 *total;
    __device__ __forceinline__ void operator()(u64 i, u32 dst) const
    {
        if (flag[i]) starts[dst] = (u32)i;
        if (i + 1 == n) *total = (u64)dst + flag[i];
    }
};

int suffix_sort_device(bwts_ctx *ctx, const u8 *d_T, u64 n, u32 **d_sa, u32 **d_rank, u32 *rounds)
{
    if (n > 0xffffffffull) return BWTS_E_RANGE;
    Alphabet al;
    BWTS_TRY(make_alphabet(ctx, d_T, n, true, &al));
    SortSpace sp;
    BWTS_TRY(sort_space_alloc(ctx, n, &sp));
    u64 active0 = 0;
    BWTS_TRY((doubling_sort<false>(ctx, d_T, n, al, nullptr, 0, sp, d_sa, rounds, &active0)));
    *d_rank = sp.rank;
    return BWTS_OK;
}

// On return *d_fstart points at k u32 factor starts placed at the arena's current base.
int lyndon_factors_device(bwts_ctx *ctx, const u8 *d_T, u64 n, u32 **d_fstart, u64 *k_out, u32 *rounds)
{
    const size_t mark = ctx->arena_off;
    u32 *sa = nullptr, *rank = nullptr;
    BWTS_TRY(suffix_sort_device(ctx, d_T, n, &sa, &rank, rounds));
    u8 *flag = arena_array<u8>(ctx, n);
    u32 *starts_tmp = arena_array<u32>(ctx, n);
    void *tmp = arena_alloc(ctx, scan_temp_bytes(n));
    if (!flag || !starts_tmp || !tmp) return BWTS_E_NOMEM;
    u64 *total = ctx->d_small + CNT_TOTAL;
    {
        SpanGuard g(ctx, BWTS_K_LYNDON, n, 8 * n);
        RankIn in{rank};
        MinFlagOut out{rank, flag};
        BWTS_TRY((device_scan<false, u32>(ctx, n, in, out, OpMin(), 0xffffffffu, tmp)));
        FlagIn fin{flag};
        StartOut sout{flag, starts_tmp, n, total};
        BWTS_TRY((device_scan<false, u32>(ctx, n, fin, sout, OpAdd(), 0u, tmp)));
    }
    BWTS_TRY(read_small(ctx, CNT_TOTAL, 1));
    const u64 k = ctx->h_small[CNT_TOTAL];
    if (k == 0 || k > n) return BWTS_E_INTERNAL;
    // move the list to the front of the arena region this call started at; everything else is released
    u32 *dst = (u32 *)(ctx->arena + mark);
    HIPC(hipMemcpyAsync(dst, starts_tmp, k * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));   // regions are disjoint: starts_tmp sits past the sort buffers
    HIPC(hipStreamSynchronize(ctx->stream));
    ctx->arena_off = mark + align_up(k * sizeof(u32), 256);
    *d_fstart = dst;
    *k_out = k;
    return BWTS_OK;
}

// ------------------------------------------------------------------------------------
// emission (mk_bwts_sa.c:172-188): P[p] = T[cprev(p)], bwts[r] = P[sa[r]]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prevsym_kernel(const u8 *__restrict__ T, u64 n, u8 *__restrict__ P)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) P[i] = i ? T[i - 1] : T[n - 1];
}
__global__ __launch_bounds__(256) void prevsym_fix_kernel(const u8 *__restrict__ T, u64 n, const u32 *__restrict__ fstart, u64 k,
                                                          u8 *__restrict__ P)
{
    const u64 f = (u64)blockIdx.x * 256 + threadIdx.x;
    if (f < k) P[fstart[f]] = T[factor_end(fstart, k, n, f) - 1];
}
__global__ __launch_bounds__(256) void emit_kernel(const u32 *__restrict__ SA, const u8 *__restrict__ P, u64 n, u8 *__restrict__ out)
{
    // 4 slots per thread: one packed 4-byte store
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 i = q * 4;
    if (i + 4 <= n && ((uintptr_t)out & 3) == 0) {
        const uint4 s = *(const uint4 *)(SA + i);
        const u32 w = (u32)P[s.x] | ((u32)P[s.y] << 8) | ((u32)P[s.z] << 16) | ((u32)P[s.w] << 24);
        *(u32 *)(out + i) = w;
    } else {
        for (u64 j = i; j < n && j < i + 4; j++) out[j] = P[SA[j]];
    }
}

size_t forward_arena_bytes(u64 n)
{
    // factor list (worst case n entries) + sort space + Lyndon temporaries (flag, starts, scan temp) + P
    return align_up(n * 4, 256) + sort_space_bytes(n) + align_up(n, 256) + align_up(n * 4, 256) + scan_temp_bytes(n) +
           align_up(n, 256) + (1 << 16);
}

int forward_device_impl(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out)
{
    if (n > 0x100000000ull) return BWTS_E_RANGE;
    BWTS_TRY(arena_reserve(ctx, forward_arena_bytes(n)));

    // 1. Lyndon factors
    u32 *d_fstart = nullptr;
    u64 k = 0;
    u32 lrounds = 0;
    BWTS_TRY(lyndon_factors_device(ctx, d_in, n, &d_fstart, &k, &lrounds));
    ctx->tm.factors = k;
    ctx->tm.lyndon_rounds = lrounds;

    // 2. cyclic sort
    Alphabet al;
    BWTS_TRY(make_alphabet(ctx, d_in, n, false, &al));
    ctx->tm.key_symbols = (u32)al.msym;
    ctx->tm.key_bits = (u32)al.key_bits;
    SortSpace sp;
    BWTS_TRY(sort_space_alloc(ctx, n, &sp));
    u32 *SA = nullptr;
    u32 rounds = 0;
    u64 active0 = 0;
    BWTS_TRY((doubling_sort<true>(ctx, d_in, n, al, d_fstart, k, sp, &SA, &rounds, &active0)));
    ctx->tm.rounds = rounds;
    ctx->tm.active_after_round0 = active0;

    // 3. emission
    u8 *P = arena_array<u8>(ctx, n);
    if (!P) return BWTS_E_NOMEM;
    {
        SpanGuard g(ctx, BWTS_K_OTHER, n, 2 * n);
        u64 blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
        prevsym_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(d_in, n, P);
        prevsym_fix_kernel<<<dim3((unsigned)((k + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_in, n, d_fstart, k, P);
    }
    {
        SpanGuard g(ctx, BWTS_K_EMIT, n, 6 * n);
        const u64 quads = (n + 3) / 4;
        emit_kernel<<<dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, ctx->stream>>>(SA, P, n, d_out);
    }
    HIPC(hipGetLastError());
    return BWTS_OK;
}
