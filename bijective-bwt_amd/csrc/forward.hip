// forward.hip -- BWTS forward transform on the GPU.
//
// Replaces divsufsort() + make_bwts_sa() + move_lyndonword_head()
// (/root/reference/mk_bwts_sa.c:48, :114-195, :74-112).  Instead of suffix-sorting and then
// patching SA/ISA sequentially, the engine
//   1. finds the Lyndon factors.  The reference reads them off the suffix array as the strict
//      prefix minima of ISA (mk_bwts_sa.c:126-129); the fast path here gets the same set without
//      a suffix sort: a device-wide prefix-min over packed m-symbol keys leaves a handful of
//      candidates, which one workgroup settles with exact suffix comparisons.  Inputs with too
//      many candidates (a^n, (ab)^n ...) take the general path: suffix sort + prefix-min of ISA;
//   2. sorts all positions directly by their infinite cyclic word rot(p)^omega with prefix
//      doubling on a CYCLIC successor (no fix-up needed: the result IS the fixed-up order);
//   3. emits bwts[r] = T[cprev(sa[r])] (mk_bwts_sa.c:172-188) as a gather.
// All index work is u32 with u64 loop bounds; bytes are unsigned.
#include "internal.h"
#include "device_utils.h"
#include "scan_templ.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

// ------------------------------------------------------------------------------------
// small-word layout in ctx->d_small / h_small (u64 words)
// ------------------------------------------------------------------------------------
#define SM_HIST      0      // 256 words: byte histogram
#define SM_CODES     256    // 32 words = 256 bytes: byte -> symbol code
#define SM_COUNTERS  320    // scratch counters
#define   CNT_ACTIVE   (SM_COUNTERS + 0)
#define   CNT_SPLITS   (SM_COUNTERS + 1)
#define   CNT_TOTAL    (SM_COUNTERS + 2)
#define   CNT_CAND     (SM_COUNTERS + 4)
#define   CNT_LYN_K    (SM_COUNTERS + 5)
#define   CNT_LYN_OVF  (SM_COUNTERS + 6)

struct Alphabet {
    int sigma;      // distinct byte values present
    int bits;       // bits per symbol code
    int msym;       // symbols packed into a round-0 key
    int key_bits;   // bits * msym (fixed-width codes) or the chosen key width (variable-length codes)
    int pad_add;    // added to a table code by the kernels (9-bit padded alphabet only)
    bool varlen;    // order-preserving variable-length codes (optimal alphabetic tree) instead of fixed-width ones
    int hstep;      // symbols every key is guaranteed to cover: the step of round 1 (msym, or key_bits / longest code)
    int patch_span; // positions in front of a factor's end whose key wraps around (msym - 1, or 64)
};
#define SM_DGCNT  2816      // 8 + 1024 words: counters of the group-local dense rounds (dense_round_kernel)
#define SM_SEGCNT 2560      // 256 words: large-group element counts of seg_small_sort_kernel (spread: one address would serialise)
#define SM_VTAB 2048        // 256 words: (length << 32) | code of each byte value (variable-length codes)
#define VL_MAXLEN 24        // longest code the variable-length key builder accepts

static int bitlen_u64(u64 x) { int b = 0; while (x) { b++; x >>= 1; } return b; }

// ------------------------------------------------------------------------------------
// byte histogram (also used by the inverse: unbwts.c:34-36)
// ------------------------------------------------------------------------------------
// 16 copies of the bins, picked by the lane id: on skewed text a quarter of a wave's lanes would otherwise hit the same
// counter in one LDS instruction and serialise
#define BH_COPIES 16
__global__ __launch_bounds__(256) void byte_hist_kernel(const u8 *__restrict__ T, u64 n, u64 *__restrict__ hist)
{
    __shared__ u32 bins[BH_COPIES][256];
    const int tid = threadIdx.x;
    u32 *mine = bins[tid & (BH_COPIES - 1)];
    for (int i = tid; i < BH_COPIES * 256; i += 256) ((u32 *)bins)[i] = 0;
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
                for (int b = 0; b < 4; b++) atomicAdd(&mine[(ws[a] >> (8 * b)) & 255u], 1u);
            }
        }
        for (u64 i = nvec * 16 + (u64)blockIdx.x * 256 + tid; i < n; i += (u64)gridDim.x * 256) atomicAdd(&mine[T[i]], 1u);
    } else {
        for (u64 i = (u64)blockIdx.x * 256 + tid; i < n; i += (u64)gridDim.x * 256) atomicAdd(&mine[T[i]], 1u);
    }
    __syncthreads();
    u32 s = 0;
#pragma unroll
    for (int c = 0; c < BH_COPIES; c++) s += bins[c][tid];
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

// builds the byte->code table from the histogram in h_small: the cyclic sort uses codes
// 0..sigma-1; the suffix (non-cyclic) sort reserves code 0 for "past the end" and uses 1..sigma
static int read_histogram(bwts_ctx *ctx, const u8 *d_T, u64 n)
{
    BWTS_TRY(byte_histogram_device(ctx, d_T, n, ctx->d_small + SM_HIST));
    return read_small(ctx, SM_HIST, 256);
}

// Symbols per round-0 key.  A key only has to separate most positions: for an i.i.d. source with collision entropy
// H2 bits per symbol, about n * 2^(log2 n - m * H2) positions stay tied after m symbols.  Taking the smallest m that
// keeps that below ~0.4 % of n saves whole radix passes on high-entropy inputs (uniform bytes: 40 instead of 64 key
// bits at 2^28); a wrong guess (real text is not i.i.d.) only moves work into the later rounds, never changes the result.
static int pick_key_symbols(const u64 *hist, u64 n, int bits, int max_sym)
{
    double sum2 = 0.0;
    for (int c = 0; c < 256; c++) { const double p = (double)hist[c] / (double)n; sum2 += p * p; }
    if (sum2 >= 1.0) return max_sym;                            // one symbol: nothing separates
    const double H2 = -log2(sum2);
    const double need = log2((double)n) + 8.0;                  // bits of collision entropy wanted in a key
    int m = (int)ceil(need / H2 - 0.05);                        // a hair of slack: exactly-uniform alphabets land on the boundary
    // whole passes are what counts: use every symbol that fits in the same number of 8-bit passes
    if (m < 1) m = 1;
    if (m > max_sym) return max_sym;
    const int passes = (m * bits + 7) / 8;
    while (m < max_sym && ((m + 1) * bits + 7) / 8 <= passes) m++;
    return m;
}

// scratch the key-width sampler may use (round-0 sort buffers, free at that point)
struct SampleScratch { const u8 *T; u64 *k[2]; u32 *v[2]; u32 *tile_hist; void *scan_temp; };
#define KEY_SAMPLES (1u << 17)
#define CNT_SAMPLE (SM_COUNTERS + 16)      // 5 words: adjacent sample keys agreeing on their first 24/32/40/48/56 bits
__global__ void sample_keys_kernel(const u8 *T, u64 n, const u64 *vtab, u32 samples, u64 *out);
__global__ void count_prefix_matches_kernel(const u64 *keys, u32 samples, unsigned long long *counters);

// Optimal alphabetic (order-preserving prefix) code for the byte histogram: the classic interval DP with Knuth's
// monotone-root bound, O(sigma^2).  Order-preserving means that comparing concatenated code words bit by bit is
// comparing the symbol strings, so packed code bits sort like the text does -- but frequent symbols take fewer bits,
// so a key of B bits separates positions about as well as B bits of entropy would.  Returns the longest code length.
static int build_alphabetic_code(const u64 *hist, u64 n, u32 *code, u8 *len, double *avg_len, double *entropy, double smooth_scale = 1.0)
{
    // (heap, not static: contexts on different host threads may be here at the same time)
    std::vector<double> Cv(256 * 256);
    std::vector<u16> Rv(256 * 256);
    double (*C)[256] = (double (*)[256])Cv.data();
    u16 (*R)[256] = (u16 (*)[256])Rv.data();
    int sym[256], sigma = 0;
    double w[256], pre[257];
    for (int c = 0; c < 256; c++) { code[c] = 0; len[c] = 0; if (hist[c]) sym[sigma++] = c; }
    // keeps rare symbols' codes short enough for the key builder; a larger scale flattens the tree (shorter longest code)
    const double smooth = ((double)(n >> 14) + 1.0) * smooth_scale;
    pre[0] = 0;
    for (int i = 0; i < sigma; i++) { w[i] = (double)hist[sym[i]] + smooth; pre[i + 1] = pre[i] + w[i]; }
    if (sigma == 1) { len[sym[0]] = 1; *avg_len = 1; *entropy = 0; return 1; }
    for (int i = 0; i < sigma; i++) { C[i][i] = 0; R[i][i] = (u16)i; }
    for (int L = 2; L <= sigma; L++) {
        for (int i = 0; i + L - 1 < sigma; i++) {
            const int j = i + L - 1;
            int lo = R[i][j - 1], hi = R[i + 1][j];
            if (lo < i) lo = i;
            if (hi > j - 1) hi = j - 1;
            if (hi < lo) hi = lo;
            double best = 1e300; int arg = lo;
            for (int k = lo; k <= hi; k++) {
                const double v = C[i][k] + C[k + 1][j];
                if (v < best) { best = v; arg = k; }
            }
            C[i][j] = best + (pre[j + 1] - pre[i]);
            R[i][j] = (u16)arg;
        }
    }
    // walk the tree: left = 0, right = 1
    struct Item { int i, j, depth; u32 prefix; } stack[512];
    int sp = 0, lmax = 0;
    stack[sp++] = Item{0, sigma - 1, 0, 0u};
    while (sp) {
        const Item it = stack[--sp];
        if (it.i == it.j) {
            len[sym[it.i]] = (u8)it.depth; code[sym[it.i]] = it.prefix;
            if (it.depth > lmax) lmax = it.depth;
            continue;
        }
        if (it.depth >= 31) return 99;
        const int k = R[it.i][it.j];
        stack[sp++] = Item{k + 1, it.j, it.depth + 1, (it.prefix << 1) | 1u};
        stack[sp++] = Item{it.i, k, it.depth + 1, it.prefix << 1};
    }
    double al = 0, h = 0;
    for (int i = 0; i < sigma; i++) {
        const double p = (double)hist[sym[i]] / (double)n;
        al += p * len[sym[i]];
        h -= p * log2(p);
    }
    *avg_len = al; *entropy = h;
    return lmax;
}

static int set_alphabet(bwts_ctx *ctx, bool reserve_pad, u64 n, Alphabet *al, const SampleScratch *ss = nullptr)
{
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
    al->pad_add = 0;
    if (bits > 8) {
        // 256 symbols + pad: 9-bit codes; the u8 table cannot hold code 256, so the kernels add
        // the +1 themselves (the table then holds 0..255)
        for (int c = 0, s = 0; c < 256; c++) if (ctx->h_small[SM_HIST + c]) codes[c] = (u8)(s++);
        al->pad_add = 1;
    }
    al->sigma = sigma;
    al->bits = bits;
    al->msym = pick_key_symbols(ctx->h_small + SM_HIST, n, bits, 64 / bits);
    const char *env = bwts_knob(ctx, "BWTS_KEY_SYMBOLS");              // tuning / test knob: force the symbol count (0 = maximum)
    if (env) { int v = atoi(env); if (v >= 1 && v <= 64 / bits) al->msym = v; else if (v == 0) al->msym = 64 / bits; }
    al->key_bits = al->bits * al->msym;
    al->varlen = false;
    al->hstep = al->msym;
    al->patch_span = al->msym - 1;
    // variable-length codes when they save whole radix passes (skewed alphabets: text); the suffix sort of the general
    // Lyndon path keeps fixed-width codes with the pad symbol
    const char *vl = bwts_knob(ctx, "BWTS_VARLEN");                    // 0 = never, 1 = always (tests), unset = when it pays
    if (!reserve_pad && sigma > 1 && !(vl && vl[0] == '0') && !(env && !vl)) {
        u32 vcode[256]; u8 vlen[256];
        double avg = 0, ent = 0;
        const int lmax = build_alphabetic_code(ctx->h_small + SM_HIST, n, vcode, vlen, &avg, &ent);
        int lmax_eff = lmax;
        if (lmax <= VL_MAXLEN) {
            // Width: the smallest whole number of bytes B for which an i.i.d. source with this histogram leaves at most
            // ~0.8 % of the positions tied.  share[b] = probability that two independent positions agree on the first b
            // bits of their code streams: both start with the same symbol and agree on the rest, or their first code
            // words are both longer than b bits and agree on those b bits.
            int kb = 64;
            {
                double p[256], share[65];
                for (int c = 0; c < 256; c++) p[c] = (double)ctx->h_small[SM_HIST + c] / (double)n;
                share[0] = 1.0;
                for (int b = 1; b <= 64; b++) {
                    double s = 0.0;
                    for (int c = 0; c < 256; c++)
                        if (vlen[c] && vlen[c] <= b) s += p[c] * p[c] * share[b - vlen[c]];
                    // code words longer than b bits that share their first b bits are neighbours in symbol order
                    // (alphabetic and prefix-free), so the partial term is a sum of squared run masses
                    double mass = 0.0; u32 cur = 0; bool open = false;
                    for (int c = 0; c < 256; c++) {
                        if (!vlen[c] || vlen[c] <= b) continue;
                        const u32 pc = vcode[c] >> (vlen[c] - b);
                        if (open && pc == cur) mass += p[c];
                        else { s += mass * mass; mass = p[c]; cur = pc; open = true; }
                    }
                    s += mass * mass;
                    share[b] = s;
                }
                for (int b = 32; b <= 64; b += 8)
                    if ((double)n * share[b] <= 1.0 / 128.0) { kb = b; break; }
                // The model knows nothing about repeated phrases.  Where the input is large enough to matter, sort the keys
                // of 2^17 sampled positions and count neighbours that agree on their first B bits: if the sample shows
                // clearly more ties than the model allows, take the empirical figure (real text wants the full 64 bits).
                if (ss && n >= (1ull << 22)) {
                    u64 *tab = ctx->h_small + SM_VTAB;
                    for (int c = 0; c < 256; c++) tab[c] = ((u64)vlen[c] << 32) | vcode[c];
                    HIPC(hipMemcpyAsync(ctx->d_small + SM_VTAB, tab, 256 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
                    HIPC(hipMemsetAsync(ctx->d_small + CNT_SAMPLE, 0, 8 * sizeof(u64), ctx->stream));
                    SpanGuard g(ctx, BWTS_K_KEYBUILD, KEY_SAMPLES, 0);
                    sample_keys_kernel<<<dim3(KEY_SAMPLES / 256), dim3(256), 0, ctx->stream>>>(ss->T, n, ctx->d_small + SM_VTAB, KEY_SAMPLES, ss->k[0]);
                    SortPlan spn;
                    spn.keys[0] = ss->k[0]; spn.keys[1] = ss->k[1];
                    spn.vals[0] = ss->v[0]; spn.vals[1] = ss->v[1];
                    spn.tile_hist = ss->tile_hist; spn.scan_temp = ss->scan_temp;
                    int sres = 0;
                    BWTS_TRY(radix_sort_pairs(ctx, spn, KEY_SAMPLES, 64, &sres));
                    count_prefix_matches_kernel<<<dim3(KEY_SAMPLES / 256), dim3(256), 0, ctx->stream>>>(
                        ss->k[sres], KEY_SAMPLES, (unsigned long long *)(ctx->d_small + CNT_SAMPLE));
                    HIPC(hipGetLastError());
                    BWTS_TRY(read_small(ctx, CNT_SAMPLE, 5));
                    const double pairs = 0.5 * (double)KEY_SAMPLES * (double)KEY_SAMPLES;
                    int kb_emp = 64;
                    for (int b = 32, w = 1; b <= 56; b += 8, w++) {
                        const double m = (double)ctx->h_small[CNT_SAMPLE + w];
                        const double partners = m >= 8.0 ? (double)n * m / pairs : (double)n * share[b];   // few hits: trust the model
                        if (1.0 - exp(-partners) <= 1.0 / 64.0) { kb_emp = b; break; }
                    }
                    if (kb_emp > kb) {
                        // Repeats, not entropy, tie these positions, and no key width separates the copies of a long repeat: the
                        // rounds on the tied list will.  Wider keys then only pay where they lengthen the first step (key bits /
                        // longest code word) -- with short code words five digits (the packed passes) already give a step
                        // of four symbols, and three fewer n-sized passes beat the few per cent of extra list elements (text 2^30,
                        // longest code 9 bits: 64 / 48 / 40 / 32 key bits = 169 / 160 / 143 / 159 ms); with long code words (real
                        // text: 205 symbols, 16 bits) every key bit counts (53.6 MiB: 16.1 / 18.5 / 20.6 / 23.3 ms).
                        // (five packed digits whatever the model asks for at this n: text 2^31 / 2^32 with 48 against 40 bits = 334 / 660
                        // against 295 / 590 ms)
                        const int keep = 40;
                        if (keep / lmax >= 4) kb = keep;
                        else {
                            kb = kb_emp;
                            // the fixed-width alternative was sized by the same model: let it use every symbol that fits
                            al->msym = 64 / bits;
                            al->key_bits = al->bits * al->msym;
                            al->hstep = al->msym;
                            al->patch_span = al->msym - 1;
                        }
                    }
                }
            }
            const char *kbe = bwts_knob(ctx, "BWTS_KEY_BITS");
            if (kbe) { int v = atoi(kbe); if (v >= 8 && v >= lmax && v <= 64) kb = v; }       // (a key must hold its first symbol whole: the first step is >= 1)
            const int passes_fixed = (al->key_bits + 7) / 8, passes_var = (kb + 7) / 8;
            // equal pass counts: the variable-length key still holds more symbols when its words are shorter on average
            if ((vl && vl[0] == '1') || passes_var < passes_fixed || (passes_var == passes_fixed && avg < (double)bits - 0.25)) {
                al->varlen = true;
                al->key_bits = kb;
                al->hstep = kb / lmax_eff < 1 ? 1 : kb / lmax_eff;
                al->patch_span = 64;
                al->msym = al->hstep;
                u64 *tab = ctx->h_small + SM_VTAB;
                for (int c = 0; c < 256; c++) tab[c] = ((u64)vlen[c] << 32) | vcode[c];
                HIPC(hipMemcpyAsync(ctx->d_small + SM_VTAB, tab, 256 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
            }
        }
    }
    memcpy(ctx->h_small + SM_CODES, codes, 256);
    HIPC(hipMemcpyAsync(ctx->d_small + SM_CODES, ctx->h_small + SM_CODES, 256, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));   // h_small is reused by later read-backs
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
// round 0: key[p] = first msym symbols of the suffix at p (zero-filled past the end), value = p
// Round-0 keys in original order.  Wide: one u64 each.  Split: when the packed sort will take them (at most 40 bits,
// radix_packed_applicable()), keybuild writes the two streams the packed passes travel with --
//   lo  u32  key bits 0..31
//   c   u16  key bits 32..39 | carried byte << 8      (c16)      or  u8: the carried byte alone (keys of <= 32 bits)
// where the carried byte of position p is T[cprev(p)] (mk_bwts_sa.c:172-188): T[p-1], fixed at factor heads afterwards.
// The first pass then reads 6 (5) bytes per element like any other pass, and no separate previous-symbol array exists.
struct KeyStore { u64 *wide; u32 *lo; u8 *c; bool c16; u8 *wprev; /* beside wide keys: the carried bytes, or null */ };
__device__ __forceinline__ void ks_store(const KeyStore &ks, u64 i, u64 key)        // the key only (patches)
{
    if (ks.wide) ks.wide[i] = key;
    else { ks.lo[i] = (u32)key; if (ks.c16) ks.c[2 * i] = (u8)(key >> 32); }
}
__device__ __forceinline__ void ks_store_with_prev(const KeyStore &ks, u64 i, u64 key, u32 prev)   // key + carried byte (keybuild)
{
    if (ks.wide) { ks.wide[i] = key; if (ks.wprev) ks.wprev[i] = (u8)prev; }
    else {
        ks.lo[i] = (u32)key;
        if (ks.c16) ((u16 *)ks.c)[i] = (u16)(((u32)(key >> 32) & 255u) | (prev << 8));
        else ks.c[i] = (u8)prev;
    }
}
__device__ __forceinline__ u64 ks_load(const KeyStore &ks, u64 i)
{
    return ks.wide ? ks.wide[i] : ((u64)ks.lo[i] | (ks.c16 ? (u64)ks.c[2 * i] << 32 : 0ull));
}
static KeyStore key_store_of(u64 *keys0, u64 n, bool split, int key_bits)
{
    KeyStore ks;
    ks.wide = split ? nullptr : keys0;
    ks.lo = (u32 *)keys0;
    ks.c = (u8 *)keys0 + align_up((size_t)n * 4, 256);
    ks.c16 = key_bits > 32;
    ks.wprev = nullptr;
    return ks;
}
// the carried byte of a factor's first position is the factor's last byte
__global__ __launch_bounds__(256) void carried_head_fix_kernel(const u8 *__restrict__ T, u64 n, const u32 *__restrict__ fstart, u64 k, KeyStore ks);

// ------------------------------------------------------------------------------------
#define KB_THREADS 256
#define KB_ITEMS   8
#define KB_TILE    (KB_THREADS * KB_ITEMS)
#define KB_HALO    64

// The value array of round 0 is the identity and is never written: the first radix pass uses the element index.
// pos0 / lim: the launch covers positions [pos0, lim) (pos0 a multiple of KB_TILE) and stores the key of position q at index
// q - pos0, tile minima at tile - pos0 / KB_TILE: the wide path (n > 2^32) materialises keys one segment at a time.
__global__ __launch_bounds__(KB_THREADS) void keybuild0_kernel(const u8 *__restrict__ T, u64 n, const u8 *__restrict__ codes_g,
                                                               int bits, int msym, int pad_add,
                                                               KeyStore keys, u64 *__restrict__ tile_min /* may be null */, u64 pos0, u64 lim)
{
    __shared__ u16 sc[KB_TILE + KB_HALO + 16];
    __shared__ u64 skey[KB_TILE + KB_TILE / 8];     // blocked -> striped transpose (one pad slot per 8)
    __shared__ u8 codes[256];
    __shared__ u64 wmin[KB_THREADS / 64];
    __shared__ __attribute__((aligned(16))) u8 sraw[KB_TILE + KB_HALO + 16];     // the tile's raw bytes (split keys: carried byte = T[q - 1])
    __shared__ u8 before_tile;

    const int tid = threadIdx.x;
    const u64 base = pos0 + (u64)blockIdx.x * KB_TILE;
    const u64 end = base + KB_TILE < lim ? base + KB_TILE : lim;
    codes[tid] = codes_g[tid];
    if (tid == 0) before_tile = base ? T[base - 1] : T[n - 1];
    __syncthreads();
    const int key_bits = bits * msym;
    const u64 mask = key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1ull);

    // symbol codes of the tile and its halo; past the end of the text the code is 0.  16 bytes per lane
    // where the text allows it (tile bases are multiples of 2048, device buffers are 16-byte aligned).
    const u32 span = (u32)(end - base) + (u32)msym;
    const bool vec_ok = ((uintptr_t)T & 15) == 0;
    for (u32 c = tid; c * 16 < span; c += KB_THREADS) {
        const u64 q0 = base + (u64)c * 16;
        if (vec_ok && q0 + 16 <= n) {
            const uint4 v = *(const uint4 *)(T + q0);
            *(uint4 *)(sraw + c * 16) = v;
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int bb = 0; bb < 4; bb++)
                    sc[c * 16 + a * 4 + bb] = (u16)((u32)codes[(w[a] >> (8 * bb)) & 255u] + (u32)pad_add);
        } else {
            for (int bb = 0; bb < 16; bb++) {
                const u64 q = q0 + bb;
                const u8 t = q < n ? T[q] : (u8)0;
                sraw[c * 16 + bb] = t;
                sc[c * 16 + bb] = q < n ? (u16)((u32)codes[t] + (u32)pad_add) : (u16)0;
            }
        }
    }
    __syncthreads();
    const u32 o = (u32)tid * KB_ITEMS;
    u64 lo = ~0ull;
    if (base + o < end) {
        u64 key = 0;
        for (int j = 0; j < msym; j++) key = (key << bits) | sc[o + j];
#pragma unroll
        for (int e = 0; e < KB_ITEMS; e++) {
            if (base + o + e < end) {
                skey[o + e + ((o + e) >> 3)] = key;
                lo = key < lo ? key : lo;
                key = ((key << bits) | sc[o + e + msym]) & mask;   // stays inside the loaded span / halo
            }
        }
    }
    if (tile_min) {
        lo = wave_scan_inclusive(lo, OpMin());
        if (lane_id() == 63) wmin[wave_id()] = lo;
    }
    __syncthreads();
    // lane-consecutive (coalesced) key stores
#pragma unroll
    for (int j = 0; j < KB_ITEMS; j++) {
        const u32 e = (u32)j * KB_THREADS + tid;
        if (base + e < end) {
            const u64 q = base + e;
            const u32 prev = e ? (u32)sraw[e - 1] : (u32)before_tile;          // T[q - 1], from the tile already in LDS
            ks_store_with_prev(keys, q - pos0, skey[e + (e >> 3)], prev);
        }
    }
    // smallest key of the tile: the Lyndon candidate search starts from these (same tile size as the scan)
    if (tile_min && tid == 0) {
        u64 t = wmin[0];
        for (int w = 1; w < KB_THREADS / 64; w++) t = wmin[w] < t ? wmin[w] : t;
        tile_min[blockIdx.x] = t;
    }
}

// ---- variable-length codes -------------------------------------------------------------------------------------
// key = the first key_bits bits of the concatenated code words starting at position q (right-aligned in the u64)
__device__ __forceinline__ u64 vl_key_plain(const u8 *__restrict__ T, u64 n, const u64 *__restrict__ vtab, int key_bits, u64 q)
{
    u64 acc = 0;
    int filled = 0;
    while (filled < key_bits) {
        if (q >= n) { acc <<= (key_bits - filled); break; }         // past the text: zero bits
        const u64 e = vtab[T[q++]];
        const int l = (int)(e >> 32), room = key_bits - filled;
        const u64 c = (u32)e;
        if (l <= room) { acc = (acc << l) | c; filled += l; }
        else { acc = (acc << room) | (c >> (l - room)); filled = key_bits; }
    }
    return acc;
}
__device__ __forceinline__ u64 vl_key_cyclic(const u8 *__restrict__ T, const u64 *__restrict__ vtab, int key_bits, u64 q, u64 s, u64 e_)
{
    u64 acc = 0;
    int filled = 0;
    while (filled < key_bits) {
        const u64 e = vtab[T[q]];
        if (++q == e_) q = s;
        const int l = (int)(e >> 32), room = key_bits - filled;
        const u64 c = (u32)e;
        if (l <= room) { acc = (acc << l) | c; filled += l; }
        else { acc = (acc << room) | (c >> (l - room)); filled = key_bits; }
    }
    return acc;
}

// same key, for the common case that 16 text bytes from q lie inside the factor: one unaligned 16-byte read, table in LDS
struct __attribute__((packed, aligned(1))) TextChunk16 { u64 w[2]; };
__device__ __forceinline__ u64 vl_key_cyclic_fast(const u8 *__restrict__ T, const u64 *vtab_lds, const u64 *__restrict__ vtab_g, int key_bits,
                                                  u64 q, u64 s, u64 e_)
{
    if (q + 16 > e_) return vl_key_cyclic(T, vtab_g, key_bits, q, s, e_);
    const TextChunk16 ch = *(const TextChunk16 *)(T + q);
    u64 acc = 0;
    int filled = 0;
#pragma unroll
    for (int b = 0; b < 16; b++) {
        if (filled < key_bits) {
            const u32 sym = (u32)((b < 8 ? ch.w[0] >> (8 * b) : ch.w[1] >> (8 * (b - 8))) & 255u);
            const u64 e = vtab_lds[sym];
            const int l = (int)(e >> 32), room = key_bits - filled;
            const u64 c = (u32)e;
            if (l <= room) { acc = (acc << l) | c; filled += l; }
            else { acc = (acc << room) | (c >> (l - room)); filled = key_bits; }
        }
    }
    if (filled < key_bits) return vl_key_cyclic(T, vtab_g, key_bits, q, s, e_);     // codes shorter than 4 bits on average: rare
    return acc;
}

__global__ __launch_bounds__(256) void sample_keys_kernel(const u8 *__restrict__ T, u64 n, const u64 *__restrict__ vtab, u32 samples,
                                                          u64 *__restrict__ out)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= samples) return;
    const u64 pos = (((u64)i * 0x9E3779B97F4A7C15ull) >> 11) % n;       // scattered, reproducible sample positions
    out[i] = vl_key_plain(T, n, vtab, 64, pos);
}
__global__ __launch_bounds__(256) void count_prefix_matches_kernel(const u64 *__restrict__ keys, u32 samples,
                                                                   unsigned long long *__restrict__ counters)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const bool ok = i > 0 && i < samples;
    const u64 a = ok ? keys[i] : 0, b = ok ? keys[i - 1] : ~0ull;
#pragma unroll
    for (int w = 0; w < 5; w++) {
        const int bits = 24 + 8 * w;
        const u64 m = __ballot(ok && (a >> (64 - bits)) == (b >> (64 - bits)));
        if (m && lane_id() == 0) atomicAdd(&counters[w], (unsigned long long)__popcll(m));
    }
}

// Round-0 keys from variable-length codes.  The tile's code words (tile + halo, KB_SYMS symbols) are laid out once as a
// bit stream in LDS -- per-thread serial prefix of the code lengths, a workgroup scan for the bit offsets, each word OR-ed
// in at its offset -- and the key of a position is the stream's first key_bits bits from that position's offset: a
// 64-bit window out of three stream words.  Symbols past the end of the text have length 0 (zero bits follow).
// (The first version kept a shifting 128-bit reservoir per thread: 118 VALU instructions per position, 91 % VALU busy.)
// Same outputs as keybuild0_kernel.
#define KB_SYMS      (KB_TILE + KB_HALO)                    // 2112
#define KB_SYM_PER   ((KB_SYMS + KB_THREADS - 1) / KB_THREADS)      // symbols a thread lays out: 9
#define KB_BS_WORDS  ((KB_SYMS * VL_MAXLEN + 31) / 32 + 4)   // stream words incl. zero padding for the last windows
__global__ __launch_bounds__(KB_THREADS) void keybuild0v_kernel(const u8 *__restrict__ T, u64 n, const u64 *__restrict__ vtab_g,
                                                                int key_bits, KeyStore keys, u64 *__restrict__ tile_min, u64 pos0, u64 lim)
{
    __shared__ u64 vtab[256];
    __shared__ __attribute__((aligned(16))) u8 sb[KB_SYMS + 16];
    __shared__ u32 bs[KB_BS_WORDS];                 // bit stream, most significant bit of a word first
    __shared__ u16 symoff[KB_SYMS + 8];             // bit offset of every symbol's code word
    __shared__ u64 wmin[KB_THREADS / 64];
    __shared__ u32 scan_sm[KB_THREADS / 64];
    __shared__ u8 before_tile;                      // T[base - 1]: the carried byte of the tile's first position

    const int tid = threadIdx.x;
    const u64 base = pos0 + (u64)blockIdx.x * KB_TILE;
    const u64 end = base + KB_TILE < lim ? base + KB_TILE : lim;
    vtab[tid] = vtab_g[tid];
    if (tid == 0) before_tile = base ? T[base - 1] : T[n - 1];
    const u32 span = KB_SYMS;
    const u64 avail = n - base;                                 // symbols of the text from the tile's start
    const u32 nvalid = avail < span ? (u32)avail : span;
    const bool vec_ok = ((uintptr_t)T & 15) == 0;
    for (u32 c = tid; c * 16 < span; c += KB_THREADS) {
        const u64 q0 = base + (u64)c * 16;
        if (vec_ok && q0 + 16 <= n) {
            *(uint4 *)(sb + c * 16) = *(const uint4 *)(T + q0);
        } else {
            for (int bb = 0; bb < 16; bb++) sb[c * 16 + bb] = q0 + bb < n ? T[q0 + bb] : (u8)0;
        }
    }
    for (u32 i = tid; i < KB_BS_WORDS; i += KB_THREADS) bs[i] = 0;
    __syncthreads();

    // lay out symbols [s0, s0 + KB_SYM_PER)
    const u32 s0 = (u32)tid * KB_SYM_PER;
    u32 ent_len[KB_SYM_PER], ent_code[KB_SYM_PER];
    u32 mine = 0;
#pragma unroll
    for (int i = 0; i < KB_SYM_PER; i++) {
        const u32 sidx = s0 + i;
        u64 ent = 0;
        if (sidx < nvalid) ent = vtab[sb[sidx]];            // past the text (or past the halo): length 0
        ent_len[i] = (u32)(ent >> 32);
        ent_code[i] = (u32)ent;
        mine += ent_len[i];
    }
    u32 total;
    u32 off = block_scan_exclusive<u32, OpAdd, KB_THREADS / 64>(mine, OpAdd(), 0u, scan_sm, &total);
#pragma unroll
    for (int i = 0; i < KB_SYM_PER; i++) {
        const u32 sidx = s0 + i;
        if (sidx < KB_SYMS) symoff[sidx] = (u16)off;
        const u32 l = ent_len[i];
        if (l) {
            const u32 wd = off >> 5, r = off & 31u;
            if (r + l <= 32) atomicOr(&bs[wd], ent_code[i] << (32 - r - l));
            else {
                atomicOr(&bs[wd], ent_code[i] >> (r + l - 32));
                atomicOr(&bs[wd + 1], ent_code[i] << (64 - r - l));
            }
        }
        off += l;
    }
    __syncthreads();

    // keys, lane-consecutive over the tile's positions (any thread can cut any position's window out of the stream):
    // coalesced stores straight from registers
    u64 lo = ~0ull;
#pragma unroll
    for (int j = 0; j < KB_ITEMS; j++) {
        const u32 e = (u32)j * KB_THREADS + tid;
        if (base + e < end) {
            const u32 bo = symoff[e];
            const u32 wd = bo >> 5, r = bo & 31u;
            const u64 hi64 = ((u64)bs[wd] << 32) | (u64)bs[wd + 1];
            const u64 win = r ? (hi64 << r) | ((u64)bs[wd + 2] >> (32 - r)) : hi64;      // 64 stream bits from bo
            const u64 key = win >> (64 - key_bits);
            lo = key < lo ? key : lo;
            const u32 prev = e ? (u32)sb[e - 1] : (u32)before_tile;            // T[q - 1], from the tile already in LDS
            ks_store_with_prev(keys, base + e - pos0, key, prev);
        }
    }
    if (tile_min) {
        lo = wave_scan_inclusive(lo, OpMin());
        if (lane_id() == 63) wmin[wave_id()] = lo;
        __syncthreads();
        // smallest key of the tile: the Lyndon candidate search starts from these (same tile size as the scan)
        if (tid == 0) {
            u64 t = wmin[0];
            for (int w = 1; w < KB_THREADS / 64; w++) t = wmin[w] < t ? wmin[w] : t;
            tile_min[blockIdx.x] = t;
        }
    }
}

// keys of the (up to 64) positions in front of each factor end wrap around inside the factor
__global__ __launch_bounds__(256) void cyclic_patch_vl_kernel(const u8 *__restrict__ T, u64 n, const u64 *__restrict__ vtab, int key_bits,
                                                              const u32 *__restrict__ fstart, u64 k, KeyStore keys)
{
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= k * 64) return;
    const u64 f = t / 64, j = t % 64;
    const u64 s = fstart[f], e = factor_end(fstart, k, n, f);
    if (j >= e - s) return;
    const u64 p = e - 1 - j;
    ks_store(keys, p, vl_key_cyclic(T, vtab, key_bits, p, s, e));
}

__global__ __launch_bounds__(256) void iota_kernel(u32 *__restrict__ v, u64 n)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) v[i] = (u32)i;
}

// first msym symbols of rot(p)^omega for a position of factor [s, e)
__device__ __forceinline__ u64 cyclic_key(const u8 *__restrict__ T, const u8 *__restrict__ codes, int bits, int msym,
                                          u64 p, u64 s, u64 e)
{
    u64 key = 0, q = p;
    for (int j = 0; j < msym; j++) {
        key = (key << bits) | (u64)codes[T[q]];
        if (++q == e) q = s;
    }
    return key;
}

// positions closer than msym to their factor's end wrap around: rewrite their round-0 keys
__global__ __launch_bounds__(256) void cyclic_patch_kernel(const u8 *__restrict__ T, u64 n, const u8 *__restrict__ codes,
                                                           int bits, int msym, const u32 *__restrict__ fstart, u64 k,
                                                           KeyStore keys)
{
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 per = (u64)(msym - 1);
    if (per == 0 || t >= k * per) return;
    const u64 f = t / per, j = t % per;
    const u64 s = fstart[f], e = factor_end(fstart, k, n, f);
    if (j >= e - s) return;
    const u64 p = e - 1 - j;
    ks_store(keys, p, cyclic_key(T, codes, bits, msym, p, s, e));
}

// ------------------------------------------------------------------------------------
// later rounds: key = (group head, rank of the h-th (cyclic) successor)
// ------------------------------------------------------------------------------------
// Sparse rank mode (few tied elements): no rank array exists.  The rank of a position that round 0 left alone is
// its slot = the number of sorted round-0 keys below its own key (binary search in the sorted key array); the rank
// of a position that round 0 left tied comes from a small map: tpos = those positions, sorted; trank = their current
// ranks, refreshed by every round.  So round 0 never scatters ranks (n random 4-byte writes), and a third or fourth
// round for a handful of stubborn ties costs a handful of binary searches, not an n-sized rank build.
// pdir (may be null): pdir[k] = first map entry whose position is >= k << psh, pdir[last + 1] = a0
__device__ __forceinline__ u64 tied_find(const u32 *__restrict__ tpos, u64 a0, u32 q, const u32 *__restrict__ pdir = nullptr, int psh = 0)
{
    u64 lo = 0, hi = a0;
    if (pdir) { const u32 k = q >> psh; lo = pdir[k]; hi = pdir[k + 1]; }
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (tpos[mid] < q) lo = mid + 1; else hi = mid;
    }
    return (lo < a0 && tpos[lo] == q) ? lo : ~0ull;
}

// cyclic successor inside the factor [s, s + L): position of the h-th symbol after p.  h < L is the common case (one factor
// holds most of a natural text), and then no division is needed: (p - s) + h < 2 L.
__device__ __forceinline__ u64 cyclic_successor(u64 p, u64 s, u64 L, u64 h)
{
    const u64 hm = h < L ? h : h % L;
    u64 off = (p - s) + hm;
    if (off >= L) off -= L;
    return s + off;
}

// Rank of a key in the sorted round-0 keys.  A plain binary search over n = 2^30 keys is 30 dependent probes, the last
// ~14 of them cache misses.  A directory over the keys' top dlog bits (dir[k] = first slot whose key's top bits are >= k,
// dir[2^dlog] = n) narrows the range to ~n / 2^dlog slots; the keys are close to uniform inside a bucket (entropy-coded
// symbols), so interpolation lands within a few dozen slots and a gallop from there stays inside two or three cache lines.
#define K0_DIR_LOG2_MAX 20
__global__ __launch_bounds__(256) void tpos_directory_kernel(const u32 *__restrict__ tpos, u64 a0, int psh, u64 buckets, u32 *__restrict__ pdir)
{
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k > buckets) return;
    if (k == buckets) { pdir[k] = (u32)a0; return; }
    const u64 want = k << psh;
    u64 lo = 0, hi = a0;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if ((u64)tpos[mid] < want) lo = mid + 1; else hi = mid;
    }
    pdir[k] = (u32)lo;
}
__global__ __launch_bounds__(256) void k0_directory_kernel(const u64 *__restrict__ K0, u64 n, int key_bits, int dlog, u64 *__restrict__ dir)
{
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k > (1ull << dlog)) return;
    if (k == (1ull << dlog)) { dir[k] = n; return; }
    const u64 want = k << (key_bits - dlog);
    u64 lo = 0, hi = n;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (K0[mid] < want) lo = mid + 1; else hi = mid;
    }
    dir[k] = lo;
}
__device__ __forceinline__ u64 k0_lower_bound(const u64 *__restrict__ K0, u64 n, u64 want, const u64 *__restrict__ dir, int dlog, int key_bits)
{
    u64 lo = 0, hi = n;
    if (dir) {
        const int sh = key_bits - dlog;
        const u64 k = want >> sh;
        lo = dir[k]; hi = dir[k + 1];
        if (lo == hi) return lo;
        u64 g = lo;
        if (sh > 0) {
            const double frac = (double)(want & ((1ull << sh) - 1ull)) / (double)(1ull << sh);
            g = lo + (u64)(frac * (double)(hi - lo));
        }
        if (g >= hi) g = hi - 1;
        if (K0[g] < want) {                     // the answer lies in (g, hi]
            lo = g + 1;
            for (u64 st = 1; lo < hi; st <<= 1) {
                const u64 p = g + st;
                if (p >= hi) break;
                if (K0[p] < want) lo = p + 1; else { hi = p; break; }
            }
        } else {                                // the answer lies in [lo, g]
            hi = g;
            for (u64 st = 1; lo < hi; st <<= 1) {
                if (st > g - lo) break;
                const u64 p = g - st;
                if (K0[p] < want) { lo = p + 1; break; }
                hi = p;
            }
        }
    }
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (K0[mid] < want) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <bool CYCLIC>
__global__ __launch_bounds__(256) void keybuild_sparse_kernel(const u32 *__restrict__ a_idx, const u32 *__restrict__ a_head, u64 a,
                                                              const u8 *__restrict__ T, u64 n, const u8 *__restrict__ codes,
                                                              int bits, int msym, int pad_add, u64 h, const u64 *__restrict__ K0, int rb,
                                                              const u32 *__restrict__ fstart, u64 k, u64 *__restrict__ keys,
                                                              const u64 *__restrict__ vtab /* variable-length codes, or null */, int key_bits,
                                                              const u32 *__restrict__ tpos, const u32 *__restrict__ trank, u64 a0,
                                                              const u64 *__restrict__ dir, int dlog,
                                                              const u32 *__restrict__ pdir, int psh)
{
    __shared__ u64 svtab[256];
    if (vtab) svtab[threadIdx.x] = vtab[threadIdx.x];
    __syncthreads();
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= a) return;
    const u64 p = a_idx[i];
    u64 q, s = 0, e = 0;
    bool past_end = false;
    if (CYCLIC) {
        const u64 f = factor_of(fstart, k, p);
        s = fstart[f]; e = factor_end(fstart, k, n, f);
        const u64 L = e - s;
        q = cyclic_successor(p, s, L, h);
    } else {
        q = p + h;
        past_end = q >= n;
    }
    u64 r = 0;
    if (!past_end) {
        const u64 j = tied_find(tpos, a0, (u32)q, pdir, psh);
        if (j != ~0ull) {
            r = trank[j];
        } else {
            u64 want;
            if (CYCLIC) {
                want = vtab ? vl_key_cyclic_fast(T, svtab, vtab, key_bits, q, s, e) : cyclic_key(T, codes, bits, msym, q, s, e);
            } else {
                want = 0;
                for (int jj = 0; jj < msym; jj++) {
                    const u64 rr = q + jj;
                    want = (want << bits) | (rr < n ? (u64)codes[T[rr]] + (u64)pad_add : 0ull);
                }
            }
            r = k0_lower_bound(K0, n, want, dir, dlog, key_bits);
        }
    }
    const u64 r2 = CYCLIC ? r : (past_end ? 0ull : r + 1ull);
    keys[i] = ((u64)a_head[i] << rb) | r2;
}

// tied map construction: keys = positions (sorted by the radix sort), values = round-0 group heads
__global__ __launch_bounds__(256) void tied_map_keys_kernel(const u32 *__restrict__ idx, u64 a, u64 *__restrict__ keys)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < a) keys[i] = idx[i];
}
__global__ __launch_bounds__(256) void tied_map_finish_kernel(const u64 *__restrict__ keys, u64 a, u32 *__restrict__ tpos)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < a) tpos[i] = (u32)keys[i];
}

// ------------------------------------------------------------------------------------
// re-rank: ONE scan over the sorted keys gives group heads and compacts the still-tied elements
// ------------------------------------------------------------------------------------
// scan element: high word = slot of a group's first element (max-scan), low word = 1 for an
// element of a group with more than one member (sum-scan)
struct OpHeadCount {
    template <typename T> __device__ __forceinline__ T operator()(T a, T b) const
    {
        const u32 ha = (u32)(a >> 32), hb = (u32)(b >> 32);
        return ((u64)(ha > hb ? ha : hb) << 32) | (u64)(u32)((u32)a + (u32)b);
    }
};

// The scan kernels visit element i = tile_base + j * 256 + tid, so the neighbouring keys K[i-1], K[i+1]
// sit in the neighbouring lanes: one global load per element; the wave's two edge lanes fetch the key across
// the wave boundary with a second, predicated load.  Loads are split from their use (load()/make()) so that a
// thread's eight loads are all in flight before the first flag is computed.
struct GroupRaw { u64 k, edge; u32 slot; };

struct GroupIn {
    const u64 *K; const u32 *S; u64 a; int rb;     // S == nullptr: slot(i) = i (round 0)
    __device__ __forceinline__ GroupRaw load(u64 i) const
    {
        GroupRaw r;
        r.k = K[i];
        const int lane = lane_id();
        const u64 act = __ballot(true);                       // lanes that hold an element (the tail tile is ragged)
        const bool next_here = lane < 63 && ((act >> ((lane + 1) & 63)) & 1ull);
        r.edge = 0;
        if (lane == 0 && i > 0) r.edge = K[i - 1];
        if (lane != 0 && !next_here && i + 1 < a) r.edge = K[i + 1];
        r.slot = S ? S[i] : (u32)i;
        return r;
    }
    __device__ __forceinline__ u64 make(u64 i, const GroupRaw &r, u32 *note) const
    {
        const u64 ki = r.k;
        const int lane = lane_id();
        const u64 act = __ballot(true);
        const bool next_here = lane < 63 && ((act >> ((lane + 1) & 63)) & 1ull);
        u64 kp = shfl_up_t(ki, 1), kn = shfl_down_t(ki, 1);
        if (lane == 0) kp = r.edge;                           // lane 0 is never also the last holder unless alone; see below
        if (!next_here) kn = r.edge;
        // a lane that is both first and last (single active lane) has only one edge slot: its successor is out of range
        // or in the next wave; that only happens in the ragged tail, where lane 0's successor is loaded here
        if (lane == 0 && !next_here && i + 1 < a) kn = K[i + 1];
        const bool f0 = i == 0 || ki != kp;
        const bool f1 = i + 1 == a || kn != ki;
        const bool so = rb >= 0 && i > 0 && (ki >> rb) == (kp >> rb);
        *note = (f0 ? 1u : 0u) | (f1 ? 2u : 0u) | (so ? 4u : 0u);
        return ((u64)(f0 ? r.slot : 0u) << 32) | (u64)((f0 && f1) ? 0u : 1u);
    }
    __device__ __forceinline__ u64 operator()(u64 i, u32 *note) const { return make(i, load(i), note); }
    __device__ __forceinline__ u64 operator()(u64 i) const { u32 note; return make(i, load(i), &note); }
};

struct GroupOut {
    const u32 *S; const u32 *V; u64 a; int rb;     // rb < 0: round 0 (no older groups)
    const u64 *K;                                   // later rounds: the sorted keys (old head = K[i] >> rb), else nullptr
    u32 *rank;                                      // dense rank array, or nullptr
    const u32 *tpos; u32 *trank; u64 a0;            // sparse rank map (tied positions -> rank), or nullptr
    u32 *SA;
    u32 *n_idx, *n_slot, *n_head;
    u64 *cnt_active, *cnt_splits;
    // note: the flags GroupIn computed for this element (same lane), so the keys are not read again
    __device__ __forceinline__ void operator()(u64 i, u64 v, u32 note) const
    {
        const bool f0 = note & 1u, f1 = note & 2u, same_old = note & 4u;
        const bool keep = !(f0 && f1);
        const u32 head = (u32)(v >> 32);
        const u32 val = (keep || rank || S) ? V[i] : 0u;     // round 0 touches the suffix array only for tied elements
        const u32 slot = S ? S[i] : (u32)i;
        // an element whose group keeps its first slot keeps its rank: no random write (dense) / map search (sparse) for it
        const bool moved = !K || (u32)(K[i] >> rb) != head;
        if (rank && moved) rank[val] = head;
        if (tpos && moved) { const u64 j = tied_find(tpos, a0, val); if (j != ~0ull) trank[j] = head; }
        if (S) SA[slot] = val;
        if (keep) {
            const u32 dst = (u32)v - 1u;
            n_idx[dst] = val; n_slot[dst] = slot; n_head[dst] = head;
        }
        // number of still-tied elements = inclusive count at the last element (no atomics on the n-sized round);
        // a count of exactly 2^32 wraps to 0 with the last element tied
        if (i + 1 == a) *cnt_active = ((u32)v == 0u && keep) ? 0x100000000ull : (u64)(u32)v;
        // "did any group split this round?" is all the driver needs: one plain store per wave, and none once the
        // flag is visibly set (counting with atomics serialised millions of waves on one address in deep rounds)
        if (rb >= 0) {
            const u64 ms = __ballot(f0 && same_old);
            if (ms && lane_id() == __ffsll((unsigned long long)__ballot(true)) - 1 &&
                __hip_atomic_load(cnt_splits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(cnt_splits, (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
};

// ---- round 0 at word granularity ---------------------------------------------------------------------------------------
// The n-sized round needs the sorted keys only to know where groups start (f0) and which elements sit in a group of more
// than one (keep).  group_flags_kernel reads the keys once and leaves both as bitmaps (one u64 per 64 slots, straight
// from __ballot); the scan then runs over n/64 words instead of n elements, and tied_from_flags_kernel turns the set
// bits into the tied list.  Same outputs as GroupIn/GroupOut with rb < 0, a third of the time.
#define GF_WORDS 8      // words (of 64 slots) a wave handles per step: eight key loads in flight per lane
// A slot starts a group when its key differs from the slot before; it is the last of its group when the NEXT slot starts one: so only
// the left neighbour's key is fetched (DPP wave_shr:1 on the two halves; lane 0 takes the word before's lane 63 from a scalar), and
// "last of its group" is the start mask shifted by one, as scalar arithmetic on the ballot words.  (The first version brought both
// neighbours through the LDS crossbar, eight ds_bpermute per element: 2.45 ms at 2^30 where the keys stream in 1.7.)
__global__ __launch_bounds__(256) void group_flags_kernel(const u64 *__restrict__ K, u64 n, u64 *__restrict__ headw, u64 *__restrict__ keepw,
                                                          u64 mask = ~0ull /* key bits that count (the 64-bit path parks position bits above them) */)
{
    const int lane = lane_id();
    const u64 words = (n + 63) / 64;
    const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6, waves = ((u64)gridDim.x * 256) >> 6;
    for (u64 w0 = wave * GF_WORDS; w0 < words; w0 += waves * GF_WORDS) {
        // (unconditional loads from clamped indices: a load under `i < n ? ... : 0` becomes a branch with a wait for the data behind it,
        // ONE load in flight per wave -- which is how this kernel ran at 3.5 TB/s for three rounds; what a slot past the end holds is
        // never looked at)
        u64 k[GF_WORDS];
#pragma unroll
        for (int q = 0; q < GF_WORDS; q++) {
            const u64 i = (w0 + q) * 64 + lane;
            k[q] = K[i < n ? i : n - 1] & mask;
        }
        const u64 first = w0 * 64, after = (w0 + GF_WORDS) * 64;          // the chunk's outer neighbours: slots first - 1 and after
        const u64 before_key = K[first > 0 ? first - 1 : 0] & mask;          // (uniform addresses: scalar loads)
        const u64 after_key = K[after < n ? after : n - 1] & mask;
        u64 hm[GF_WORDS], vm[GF_WORDS];        // per word: slots that start a group (slots past the end count as starts), valid slots
        u64 prev63 = before_key;
#pragma unroll
        for (int q = 0; q < GF_WORDS; q++) {
            const u64 i = (w0 + q) * 64 + lane;
            const u32 lo = (u32)k[q], hi = (u32)(k[q] >> 32);
            const u32 plo = (u32)__builtin_amdgcn_update_dpp((int)lo, (int)lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const u32 phi = (u32)__builtin_amdgcn_update_dpp((int)hi, (int)hi, 0x138, 0xf, 0xf, false);
            const bool valid = i < n;
            const bool diff = lane == 0 ? (i == 0 || k[q] != prev63) : (lo != plo || hi != phi);
            vm[q] = __ballot(valid);
            hm[q] = __ballot(!valid || diff);
            prev63 = ((u64)(u32)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)lo, 63);
        }
        const bool after_starts = after >= n || after_key != prev63;          // does slot `after` start a group (or lie past the end)?
        u64 mine_h = 0, mine_k = 0;
#pragma unroll
        for (int q = 0; q < GF_WORDS; q++) {
            const u64 nextbit = q + 1 < GF_WORDS ? (hm[q + 1 < GF_WORDS ? q + 1 : q] & 1ull) : (after_starts ? 1ull : 0ull);
            const u64 lastm = (hm[q] >> 1) | (nextbit << 63);                   // slots whose successor starts a group
            const u64 h = hm[q] & vm[q];
            if (lane == q) { mine_h = h; mine_k = vm[q] & ~(h & lastm); }
        }
        if (lane < GF_WORDS && w0 + lane < words) { headw[w0 + lane] = mine_h; keepw[w0 + lane] = mine_k; }
    }
}

// scan element of a word: (slot of its last group start, 0 if none) : (number of tied slots)
struct WordIn {
    const u64 *headw, *keepw;
    __device__ __forceinline__ u64 operator()(u64 w) const
    {
        const u64 hm = headw[w], km = keepw[w];
        const u64 pos = hm ? w * 64 + (u64)(63 - __clzll((long long)hm)) : 0ull;
        return (pos << 32) | (u64)__popcll(km);
    }
};

__global__ __launch_bounds__(256) void tied_from_flags_kernel(const u64 *__restrict__ headw, const u64 *__restrict__ keepw,
                                                              const u64 *__restrict__ pre, u64 n, const u32 *__restrict__ V,
                                                              u32 *__restrict__ n_idx, u32 *__restrict__ n_slot, u32 *__restrict__ n_head,
                                                              u64 *__restrict__ cnt_active)
{
    const int lane = lane_id();
    const u64 words = (n + 63) / 64;
    const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6, waves = ((u64)gridDim.x * 256) >> 6;
    for (u64 wb = wave * 64; wb < words; wb += waves * 64) {
        // lane l holds word wb + l; words with tied slots are then expanded one at a time by the whole wave
        const u64 w = wb + lane;
        u64 hm = 0, km = 0, pr = 0;
        if (w < words) { hm = headw[w]; km = keepw[w]; pr = pre[w]; }
        if (w + 1 == words) *cnt_active = (u64)(u32)pr + (u64)__popcll(km);
        u64 todo = __ballot(km != 0);
        // four words per step, their positions fetched together (whole words, from indices clamped into the array): a word's stores wait
        // for its load, so taken one at a time the words went by at one load latency each
        while (todo) {
            int q[4];
            u32 v[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                q[t] = todo ? __ffsll((unsigned long long)todo) - 1 : -1;
                todo &= todo - 1;                                   // (0 stays 0)
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const u64 i = (wb + (u64)(q[t] < 0 ? 0 : q[t])) * 64 + lane;
                v[t] = V[i < n ? i : n - 1];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (q[t] < 0) continue;
                const u64 h = shfl_t(hm, q[t]), kk = shfl_t(km, q[t]), pq = shfl_t(pr, q[t]);
                if ((kk >> lane) & 1ull) {
                    const u64 i = (wb + q[t]) * 64 + lane;
                    const u64 below = lane == 63 ? h : h & ((2ull << lane) - 1ull);        // group starts at or before this slot
                    const u32 head = below ? (u32)((wb + q[t]) * 64 + (u64)(63 - __clzll((long long)below))) : (u32)(pq >> 32);
                    const u32 dst = (u32)pq + (u32)__popcll(kk & lanemask_lt());
                    n_idx[dst] = v[t]; n_slot[dst] = (u32)i; n_head[dst] = head;
                }
            }
        }
    }
}

// rank[sa[i]] = i for every slot, then rank[idx] = head for the still-tied elements
__global__ __launch_bounds__(256) void rank_from_sa_kernel(const u32 *__restrict__ SA, u64 n, u32 *__restrict__ rank)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) rank[SA[i]] = (u32)i;
}
__global__ __launch_bounds__(256) void rank_from_list_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ head, u64 a,
                                                             u32 *__restrict__ rank)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < a) rank[idx[i]] = head[i];
}

// ------------------------------------------------------------------------------------
// the doubling sort
// ------------------------------------------------------------------------------------
struct SortSpace {
    u64 *keys[2];     // n each
    u32 *vals[2];     // n each
    u32 *rank;        // n
    u32 *tile_hist;
    void *scan_temp;
    // emission riding on round 0 (cyclic sort only): P[p] = T[cprev(p)] travels with the pairs and the last
    // pass writes it straight into the output; null = gather after the sort instead
    const u8 *carry_src = nullptr;
    u8 *carry_buf[2] = {nullptr, nullptr};
    u8 *carry_out = nullptr;
    // the round-0 tie list (slots), for patching the carried bytes of elements that later rounds reorder
    const u32 *tie_slots = nullptr;
    u64 tie_count = 0;
    // the cyclic sort will carry the byte stream (=> the packed passes may apply); keys[0] currently holds split keys
    bool want_split = false, split_keys = false;
    // set by the sort when the later rounds wrote the bytes of the tied elements themselves (dense rounds)
    bool ties_emitted = false;
};

static size_t sort_space_bytes(u64 n)
{
    return 2 * align_up(n * 8, 256) + 2 * align_up(n * 4, 256) + radix_tile_hist_bytes(n) + scan_temp_bytes(n) + 4096;
}

// the dense rank array (4 n bytes) exists only for inputs that need it: many ties after round 0, a suffix array with ranks, or a
// sort without the carried-byte buffers (whose space the round-0 flag words otherwise use)
static int ensure_rank(bwts_ctx *ctx, SortSpace &sp, u64 n)
{
    if (sp.rank) return BWTS_OK;
    char *p = nullptr;
    BWTS_TRY(aux_reserve_slot(ctx, 4, align_up((size_t)n * 4, 256), &p));
    sp.rank = (u32 *)p;
    return BWTS_OK;
}

static int sort_space_alloc(bwts_ctx *ctx, u64 n, SortSpace *sp)
{
    sp->keys[0] = arena_array<u64>(ctx, n);
    sp->keys[1] = arena_array<u64>(ctx, n);
    sp->vals[0] = arena_array<u32>(ctx, n);
    sp->vals[1] = arena_array<u32>(ctx, n);
    sp->rank = nullptr;            // ensure_rank()
    sp->tile_hist = (u32 *)arena_alloc(ctx, radix_tile_hist_bytes(n));
    sp->scan_temp = arena_alloc(ctx, scan_temp_bytes(n));
    if (!sp->keys[0] || !sp->keys[1] || !sp->vals[0] || !sp->vals[1] || !sp->tile_hist || !sp->scan_temp)
        return BWTS_E_NOMEM;
    return BWTS_OK;
}

// ---- dense rank array after round 0, binned -------------------------------------------------------------------------
// rank[SA[i]] = head(i) is n random 4-byte writes: 24 G/s on this chip, 73-85 G/s when the writes in flight fall into a
// window of <= 1 MB (tools/micro/window_scatter.hip).  So: (1) partition the pairs (SA[i], head(i)) by destination window
// -- one tile of RB_CHUNK pairs per workgroup, LDS counts, one global atomic per touched window (counters on separate
// cache lines), the tile ordered by window in LDS and written out in contiguous runs; (2) scatter window after window,
// all workgroups in the same few windows at a time.  head(i) comes straight from the round-0 flag words, so the tied
// elements need no second scatter.  SA is a permutation: every window receives exactly its size.
#define RB_THREADS 512
#define RB_PER_THREAD 16
#define RB_CHUNK (RB_THREADS * RB_PER_THREAD)
#define RB_MAX_WINDOWS 4096
#define RB_FILL_STRIDE 32
#define RB_BLOCKS_PER_WINDOW 64
static inline size_t rank_partition_lds_bytes(u32 nw) { return (size_t)RB_CHUNK * 8 + 2 * (size_t)nw * 4; }

__global__ __launch_bounds__(RB_THREADS) void rank_partition_kernel(const u32 *__restrict__ SA, u64 n, const u64 *__restrict__ headw,
                                                                    const u64 *__restrict__ pre, int wlog, u32 nw,
                                                                    u32 *__restrict__ win_fill, u64 *__restrict__ pairs)
{
    extern __shared__ __attribute__((aligned(16))) char rb_lds[];
    u64 *sorted = (u64 *)rb_lds;                       // RB_CHUNK pairs ordered by window
    u32 *cnt = (u32 *)(sorted + RB_CHUNK);             // per window: count, then the fill cursor within `sorted`
    u32 *delta = cnt + nw;                             // per window: (reserved offset in the window) - (start in `sorted`)
    __shared__ u32 scan_sm[RB_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const u64 base = (u64)blockIdx.x * RB_CHUNK;
    const u32 clen = n - base < RB_CHUNK ? (u32)(n - base) : (u32)RB_CHUNK;
    u32 x[RB_PER_THREAD], h[RB_PER_THREAD];
#pragma unroll
    for (int q = 0; q < RB_PER_THREAD; q++) {
        const u32 e = (u32)q * RB_THREADS + tid;
        x[q] = e < clen ? SA[base + e] : 0u;
    }
#pragma unroll
    for (int q = 0; q < RB_PER_THREAD; q++) {
        // slot i = base + e; its 64-slot word is the same for the whole wave
        const u64 i = base + (u32)q * RB_THREADS + tid;
        const u64 w = i >> 6;
        u64 hm = 0, pr = 0;
        if (i < n) { hm = headw[w]; pr = pre[w]; }
        const u64 below = lane == 63 ? hm : hm & ((2ull << lane) - 1ull);
        h[q] = below ? (u32)((w << 6) + (u64)(63 - __clzll((long long)below))) : (u32)(pr >> 32);
    }
    for (u32 b = tid; b < nw; b += RB_THREADS) cnt[b] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RB_PER_THREAD; q++)
        if ((u32)q * RB_THREADS + tid < clen) atomicAdd(&cnt[x[q] >> wlog], 1u);
    __syncthreads();
    {
        const u32 per = (nw + RB_THREADS - 1) / RB_THREADS;      // windows a thread scans (<= 8)
        u32 c[8], run = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 b = (u32)tid * per + j;
            c[j] = (u32)j < per && b < nw ? cnt[b] : 0u;
            run += c[j];
        }
        u32 tot;
        u32 exc = block_scan_exclusive<u32, OpAdd, RB_THREADS / 64>(run, OpAdd(), 0u, scan_sm, &tot);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 b = (u32)tid * per + j;
            if ((u32)j < per && b < nw) {
                const u32 g = c[j] ? atomicAdd(&win_fill[(size_t)b * RB_FILL_STRIDE], c[j]) : 0u;
                cnt[b] = exc;
                delta[b] = g - exc;
                exc += c[j];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RB_PER_THREAD; q++)
        if ((u32)q * RB_THREADS + tid < clen) ((uint2 *)sorted)[atomicAdd(&cnt[x[q] >> wlog], 1u)] = make_uint2(x[q], h[q]);
    __syncthreads();
    for (u32 e = tid; e < clen; e += RB_THREADS) {
        const uint2 v = ((const uint2 *)sorted)[e];
        const u32 b = v.x >> wlog;
        ((uint2 *)pairs)[((u64)b << wlog) + (u32)(delta[b] + e)] = v;
    }
}

// window b is written by RB_BLOCKS_PER_WINDOW consecutive workgroups, so the workgroups in flight share a few windows
__global__ __launch_bounds__(256) void rank_scatter_pairs_kernel(const u64 *__restrict__ pairs, const u32 *__restrict__ win_fill, int wlog,
                                                                 u32 *__restrict__ rank)
{
    const u32 b = blockIdx.x / RB_BLOCKS_PER_WINDOW, s = blockIdx.x % RB_BLOCKS_PER_WINDOW;
    const u32 fill = win_fill[(size_t)b * RB_FILL_STRIDE];
    const u32 per = (fill + RB_BLOCKS_PER_WINDOW - 1) / RB_BLOCKS_PER_WINDOW;
    const u32 lo = s * per, hi = lo + per < fill ? lo + per : fill;
    const uint2 *src = (const uint2 *)pairs + ((u64)b << wlog);
    for (u32 e = lo + threadIdx.x; e < hi; e += 256) { const uint2 v = src[e]; rank[v.x] = v.y; }
}

// Second form of the same build, the one in use: one u64 per slot, (head << 32) | position, sorted on the position's top 16 bits
// by two keys-only radix passes (16 bytes per element and pass, ballot-ranked like every other pass -- the LDS atomics of
// rank_partition_kernel made that single pass cost 30 ms at n = 2^30), then rank[position] = head with all writes of a
// workgroup inside a window of n / 2^16 positions.
__global__ __launch_bounds__(256) void rank_keys_kernel(const u32 *__restrict__ SA, u64 n, const u64 *__restrict__ headw,
                                                        const u64 *__restrict__ pre, u64 *__restrict__ keys)
{
    const int lane = lane_id();
    // four 64-slot words per wave and step, their loads issued together (from clamped indices: see DESIGN.md section 10 on loads
    // behind a bounds test): one at a time this kernel ran at 3.3 TB/s
    const u64 words = (n + 63) >> 6;
    const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6, waves = ((u64)gridDim.x * 256) >> 6;
    for (u64 w0 = wave * 4; w0 < words; w0 += waves * 4) {
        u64 hm[4], pr[4];
        u32 sa[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u64 w = w0 + q < words ? w0 + q : words - 1;
            const u64 i = (w << 6) + lane;
            hm[q] = headw[w]; pr[q] = pre[w];
            sa[q] = SA[i < n ? i : n - 1];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u64 w = w0 + q, i = (w << 6) + lane;
            if (w >= words || i >= n) continue;
            const u64 below = lane == 63 ? hm[q] : hm[q] & ((2ull << lane) - 1ull);
            const u32 h = below ? (u32)((w << 6) + (u64)(63 - __clzll((long long)below))) : (u32)(pr[q] >> 32);
            keys[i] = ((u64)h << 32) | (u64)sa[q];
        }
    }
}
// The positions are a permutation, so after the sort window w (positions [w << wlog, (w + 1) << wlog)) occupies exactly the
// keys [w << wlog, ...): one workgroup takes one window, scatters the heads into an LDS image of the window's ranks and
// writes the image out whole -- a streaming kernel (20 ms -> 3.5 ms at n = 2^30 against scattering to memory, where the four
// workgroups sharing a window sat on different XCDs and every rank line left their L2s in pieces).
#define RA_WLOG_MAX 14
__global__ __launch_bounds__(1024) void rank_apply_kernel(const u64 *__restrict__ keys, u64 n, int wlog, u32 *__restrict__ rank)
{
    extern __shared__ __attribute__((aligned(16))) u32 ra_img[];
    const u64 lo = (u64)blockIdx.x << wlog;
    const u64 hi = lo + (1ull << wlog) < n ? lo + (1ull << wlog) : n;
    for (u64 i = lo + threadIdx.x; i < hi; i += 1024) { const u64 kv = keys[i]; ra_img[(u32)kv - (u32)lo] = (u32)(kv >> 32); }
    __syncthreads();
    for (u64 i = lo + threadIdx.x; i < hi; i += 1024) rank[i] = ra_img[i - lo];
}
__global__ void count_tied_kernel(const u64 *__restrict__ keepw, const u64 *__restrict__ pre, u64 words, u64 *__restrict__ cnt_active)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *cnt_active = (u64)(u32)pre[words - 1] + (u64)__popcll(keepw[words - 1]);
}

// ---- later rounds: sorting inside groups ------------------------------------------------------------------------------
// The tied list is ordered by group (heads increase along it), so a round's sort by (head, successor rank) only has to
// order each group by the rank.  On real text most tied elements sit in small groups (53 MiB of source text: 62 % of all
// element-rounds in groups of <= 8, the late rounds almost entirely pairs), yet a radix sort of the whole list costs 7 passes.
// seg_small_sort_kernel finds every element's group extent from the list's head array; the members of a group of at most
// SEG_CAP sort themselves in place (each counts the members ordering before it); elements of larger groups are flagged and counted.  If they are the majority
// the whole list goes through the radix sort as before; otherwise only they are compacted (scan), radix-sorted
// and written back to the slots they came from -- the compaction keeps list order, and a sort by (head, rank) keeps groups
// in list order, so the j-th sorted element belongs to the j-th flagged slot.
#define SEG_CAP 8
#define SEG_HALO 8                      // list elements a wave looks at in front of the ones it owns
#define SEG_OWN  (64 - SEG_HALO - SEG_CAP - 1)      // elements a wave decides: their group's start and end lie inside its 64-lane window
// A wave looks at 64 consecutive list elements (one coalesced load of their group heads), finds group starts with one
// __ballot, and decides the SEG_OWN elements in the middle of its window: an element's group start is the highest start
// bit at or below its lane, the group's end the next start bit above that.  Elements of groups larger than SEG_CAP get
// big[i] = 1 and are counted (one atomic per wave, spread over 256 counters).  A group of 2 .. SEG_CAP whose first element
// the wave owns is sorted by its own lanes: each loads its (key, value), counts with SEG_CAP shuffles how many members order
// before it (ties by list order: stable), and stores itself to that slot of the group -- coalesced loads, near-coalesced stores.
__global__ __launch_bounds__(256) void seg_small_sort_kernel(const u32 *__restrict__ head, u64 *__restrict__ K, u32 *__restrict__ V, u64 a,
                                                             u8 *__restrict__ big, u64 *__restrict__ big_count)
{
    const int lane = lane_id();
    const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const u64 own0 = wave * SEG_OWN;                 // first element this wave decides
    if (own0 >= a) return;
    const long long e = (long long)own0 - SEG_HALO + lane;        // list element of this lane (may lie outside [0, a))
    const bool inside = e >= 0 && (u64)e < a;
    const u32 h = inside ? head[e] : 0u;
    u32 hp = (u32)__shfl_up((int)h, 1, 64);
    if (lane == 0) hp = (e > 0 && (u64)(e - 1) < a) ? head[e - 1] : 0u;
    // a start: first element of the list, or head differs from the predecessor's; elements outside the list count as
    // starts so that a group never extends over them
    const bool start = !inside || e == 0 || h != hp;
    const u64 starts = __ballot(start);
    // every lane: where does my group start and end inside the window?  (-1 / size 0 = cannot tell from here)
    const u64 below = lane == 63 ? starts : starts & ((2ull << lane) - 1ull);
    const int s_lane = below ? 63 - __clzll((long long)below) : -1;
    u32 sz = 0;
    bool too_big = s_lane < 0 || lane - s_lane >= SEG_CAP;         // start further back than SEG_CAP - 1 elements
    if (!too_big) {
        const u64 above = s_lane == 63 ? 0ull : starts >> (s_lane + 1);
        const int e_lane = above ? s_lane + 1 + (__ffsll((unsigned long long)above) - 1) : 64;
        sz = (u32)(e_lane - s_lane);
        if (sz > SEG_CAP) too_big = true;
    }
    const bool owned = inside && lane >= SEG_HALO && lane < SEG_HALO + SEG_OWN;
    if (owned) big[e] = too_big ? 1 : 0;
    const u64 bm = __ballot(owned && too_big);
    if (bm && lane == 0) atomicAdd((unsigned long long *)&big_count[wave & 255], (unsigned long long)__popcll(bm));
    // members of a small group whose first element this wave owns sort themselves
    const bool member = inside && !too_big && sz >= 2 && s_lane >= SEG_HALO && s_lane < SEG_HALO + SEG_OWN;
    if (__ballot(member) == 0) return;
    u64 k = 0;
    u32 v = 0;
    if (member) { k = K[e]; v = V[e]; }
    const u32 o = member ? (u32)(lane - s_lane) : 0u;
    u32 rank = 0;
#pragma unroll
    for (int j = 0; j < SEG_CAP; j++) {
        const int src = (s_lane < 0 ? 0 : s_lane) + j;
        const u64 ko = shfl_t(k, src & 63);
        if (member && (u32)j < sz && (ko < k || (ko == k && (u32)j < o))) rank++;
    }
    if (member) { const u64 dst = (u64)e - o + rank; K[dst] = k; V[dst] = v; }
}
// compaction of the larger groups' elements.  Scan value: low word = flagged elements, high word = flagged groups (their
// first elements) before the position.  The compacted key carries the group's ordinal among the flagged groups instead of
// its head slot -- at most m/(SEG_CAP+1) groups, so the radix sort of the compacted list needs fewer key bits (usually a
// whole pass less); seg_writeback_kernel puts the head back.
struct SegFlagIn {
    const u8 *big; const u32 *head;
    __device__ __forceinline__ u64 operator()(u64 i) const
    {
        const u32 f = big[i];
        const u32 st = f && (i == 0 || head[i] != head[i - 1]) ? 1u : 0u;
        return ((u64)st << 32) | f;
    }
};
struct SegBigOut {
    const u8 *big; const u32 *head; const u64 *K; const u32 *V; u64 a; int rb; u64 *bk; u32 *bv; u32 *bpos; u64 *count;
    __device__ __forceinline__ void operator()(u64 i, u64 before) const
    {
        const u32 f = big[i];
        if (f) {
            const u32 st = (i == 0 || head[i] != head[i - 1]) ? 1u : 0u;
            const u64 ord = (before >> 32) + st - 1;                     // groups flagged up to and including this element's, minus one
            const u32 at = (u32)before;
            bk[at] = (ord << rb) | (K[i] & ((1ull << rb) - 1ull));
            bv[at] = V[i];
            bpos[at] = (u32)i;
        }
        if (i + 1 == a) *count = (u64)(u32)before + f;
    }
};
__global__ __launch_bounds__(256) void seg_writeback_kernel(const u64 *__restrict__ sk, const u32 *__restrict__ sv, const u32 *__restrict__ bpos,
                                                            u64 m, const u32 *__restrict__ head, int rb, u64 *__restrict__ K, u32 *__restrict__ V)
{
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j < m) {
        const u32 p = bpos[j];
        K[p] = ((u64)head[p] << rb) | (sk[j] & ((1ull << rb) - 1ull));       // the slot's group (sorting keeps groups in place)
        V[p] = sv[j];
    }
}

#include "dense_rounds.h"
#include "chunk_rounds.h"

// Sorts all positions by their (cyclic | suffix) word.  sp.keys[0]/sp.vals[0] hold the round-0
// keys and the identity on entry.  want_ranks: leave final ranks in sp.rank (ISA for the suffix sort).
template <bool CYCLIC>
static int doubling_sort(bwts_ctx *ctx, const u8 *d_T, u64 n, const Alphabet &al, const u32 *d_fstart, u64 k,
                         SortSpace &sp, bool want_ranks, u32 **sa_out, u32 *rounds_out, u64 *active0_out)
{
    const u8 *d_codes = (const u8 *)(ctx->d_small + SM_CODES);
    u64 *cnt = ctx->d_small + SM_COUNTERS;

    // ---- round 0 ------------------------------------------------------------------
    SortPlan plan;
    plan.keys[0] = sp.keys[0]; plan.keys[1] = sp.keys[1];
    plan.vals[0] = sp.vals[0]; plan.vals[1] = sp.vals[1];
    plan.tile_hist = sp.tile_hist; plan.scan_temp = sp.scan_temp;
    plan.sym_src = sp.carry_src; plan.sym_buf[0] = sp.carry_buf[0]; plan.sym_buf[1] = sp.carry_buf[1]; plan.sym_final = sp.carry_out;
    plan.vals_identity = radix_supports_sym(ctx);     // keybuild0 writes no value array
    plan.keys_split = CYCLIC && sp.split_keys && plan.sym_final && plan.vals_identity;
    if (CYCLIC && sp.split_keys && !plan.keys_split) return BWTS_E_INTERNAL;      // keybuild split the keys for a sort that cannot take them
    if (!plan.vals_identity) {                     // tuning configs without the identity variant: materialise it
        u64 blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
        iota_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(sp.vals[0], n);
    }
    int res = 0;
    STAGE("cyclic patch (before round 0)");
    BWTS_TRY(radix_sort_pairs(ctx, plan, n, al.key_bits, &res));
    STAGE("round-0 sort");
    u64 *K0 = sp.keys[res];
    u32 *SA = sp.vals[res];
    // the other key buffer (8n bytes) and value buffer (4n) are free: first active list goes there
    ActiveList cur;
    cur.idx = (u32 *)sp.keys[res ^ 1];
    cur.slot = cur.idx + n;
    cur.head = sp.vals[res ^ 1];

    HIPC(hipMemsetAsync(cnt, 0, 4 * sizeof(u64), ctx->stream));
    const u64 *flag_heads = nullptr, *flag_pre = nullptr;     // set when the round-0 flag words survive outside sp.rank
    const u64 *flag_heads_any = nullptr, *flag_pre_any = nullptr, *flag_keep = nullptr;   // the flag words wherever they live
    u64 flag_words = 0;
    bool rank_early = false;                                  // the dense rank array was built before the tied list
    const bool scan_by_keys = [ctx] { const char *e = bwts_knob(ctx, "BWTS_GROUPSCAN"); return e && !strcmp(e, "keys"); }();
    if (scan_by_keys) {             // the element-wise scan the later rounds use (kept selectable for tests)
        SpanGuard g(ctx, BWTS_K_RERANK, n, 8 * n);
        GroupIn in{K0, nullptr, n, -1};
        GroupOut out{nullptr, SA, n, -1, nullptr, nullptr, nullptr, nullptr, 0, SA, cur.idx, cur.slot, cur.head, cnt + 0, cnt + 1};
        BWTS_TRY((device_scan<true, u64>(ctx, n, in, out, OpHeadCount(), (u64)0, sp.scan_temp)));
    } else {
        SpanGuard g(ctx, BWTS_K_RERANK, n, 8 * n);
        // flags and word prefixes (3 * n/8 bytes): in the carried-byte ping-pong buffers when there are any (free once the
        // sort is done), else in sp.rank, which is not needed before the rounds that follow
        const u64 words = (n + 63) / 64;
        u64 *headw, *keepw, *pre;
        if (sp.carry_buf[0] && sp.carry_buf[1] && n >= 4096) {
            headw = (u64 *)sp.carry_buf[0]; keepw = headw + words; pre = (u64 *)sp.carry_buf[1];
            flag_heads = headw; flag_pre = pre;
        } else {
            BWTS_TRY(ensure_rank(ctx, sp, n));
            headw = (u64 *)sp.rank; keepw = headw + words; pre = keepw + words;
        }
        u64 waves = (words + GF_WORDS - 1) / GF_WORDS;
        unsigned blocks = (unsigned)((waves + 3) / 4 < 16384 ? (waves + 3) / 4 : 16384);
        group_flags_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(K0, n, headw, keepw);
        WordIn in{headw, keepw};
        ScanStoreArr<u64> out{pre};
        BWTS_TRY((device_scan<false, u64>(ctx, words, in, out, OpHeadCount(), (u64)0, sp.scan_temp)));
        count_tied_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(keepw, pre, words, cnt + 0);
        HIPC(hipGetLastError());
        STAGE("group flags + word scan + count");
        flag_keep = keepw; flag_words = words;
        flag_heads_any = headw; flag_pre_any = pre;
    }
    BWTS_TRY(read_small(ctx, SM_COUNTERS, 4));
    u64 a = ctx->h_small[CNT_ACTIVE];
    *active0_out = a;
    if (CYCLIC) ctx->tm.round_active[0] = a;
    bool rank_valid = false;
    if (flag_keep) {
        // Many ties: the dense rank array.  It is built BEFORE the tied list, while the list's future home (the other key
        // buffer) is still free: the build sorts one u64 per slot between that buffer and the sorted keys' (not needed any
        // more without the sparse rank map).
        const bool plain_build = [ctx] { const char *e = bwts_knob(ctx, "BWTS_RANKBUILD"); return e && !strcmp(e, "plain"); }();
        const bool part_build = [ctx] { const char *e = bwts_knob(ctx, "BWTS_RANKBUILD"); return e && !strcmp(e, "partition"); }();
        if (a > n / 32 && flag_heads && n >= (1ull << 22) && !plain_build && !part_build) {
            BWTS_TRY(ensure_rank(ctx, sp, n));
            SpanGuard g(ctx, BWTS_K_RERANK, n, 28 * n);
            u64 *rk[2] = {sp.keys[res ^ 1], K0};
            u64 blocks = (n + 255) / 256; if (blocks > 16384) blocks = 16384;
            rank_keys_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(SA, n, flag_heads, flag_pre, rk[0]);
            HIPC(hipGetLastError());
            // sorted on the position's top 16 bits (24 beyond n = 2^30): windows of at most 2^RA_WLOG_MAX positions
            const int pb = bitlen_u64(n - 1);
            const int sbits = pb - 16 <= RA_WLOG_MAX ? 16 : 24;
            int rres = 0;
            BWTS_TRY(radix_sort_keys(ctx, rk, sp.tile_hist, sp.scan_temp, n, pb - sbits, sbits, &rres));
            const int wlog = RA_WLOG_MAX;                   // sorted on at least the bits above 2^14: every 2^14 keys are 2^14 consecutive positions
            BWTS_TRY(ensure_dyn_lds(ctx, (const void *)rank_apply_kernel, (size_t)4 << RA_WLOG_MAX));
            rank_apply_kernel<<<dim3((unsigned)((n + (1ull << wlog) - 1) >> wlog)), dim3(1024), (size_t)4 << wlog, ctx->stream>>>(rk[rres], n, wlog, sp.rank);
            HIPC(hipGetLastError());
            rank_valid = true;
            rank_early = true;
        }
        SpanGuard g(ctx, BWTS_K_RERANK, a, 12 * a);
        const u64 waves = (flag_words + 63) / 64;
        const unsigned blocks = (unsigned)((waves + 3) / 4 < 16384 ? (waves + 3) / 4 : 16384);
        tied_from_flags_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(flag_heads_any, flag_keep, flag_pre_any, n, SA, cur.idx, cur.slot, cur.head, cnt + 0);
        HIPC(hipGetLastError());
        STAGE("tied list");
    }
    sp.tie_slots = cur.slot;        // stays untouched by the later rounds
    sp.tie_count = a;
    u32 rounds = 1;
    ActiveList none{nullptr, nullptr, nullptr};

    if (a > 0) {
        // every position tied at n = 2^32 (a constant or a two-symbol periodic input of exactly 4 GiB): the side buffers of the
        // later rounds are sized by the tied count and do not fit next to the 36 n bytes of round 0
        if (a > 0xffffffffull) return BWTS_E_NOMEM;
        // few tied elements: sparse rank map; many (real text ties most m-grams): the dense rank array
        const bool sparse = a <= n / 32;
        // aux: two key buffers, one value scratch, two list sets (the group-local dense rounds lay out their own, smaller block)
        char *base = nullptr;
        const size_t e4 = align_up((size_t)a * 4, 256), e8 = align_up((size_t)a * 8, 256);
        const size_t dir_bytes = align_up(((size_t)1 << K0_DIR_LOG2_MAX) * 8 + 8, 256) + align_up(((size_t)1 << K0_DIR_LOG2_MAX) * 4 + 8, 256);
        const size_t e1 = align_up((size_t)a, 256);
        const size_t seg_bytes = e1 + 2 * e8 + 3 * e4;            // flags, compacted keys x2, values x2, slots of the larger groups
        if (sparse) BWTS_TRY(aux_reserve(ctx, 2 * e8 + 9 * e4 + dir_bytes + seg_bytes, &base));
        u64 *akeys[2] = {(u64 *)base, (u64 *)(base + e8)};
        char *q = base + 2 * e8;
        u32 *scratch = (u32 *)q; q += e4;
        ActiveList sets[2];
        for (int s = 0; s < 2; s++) {
            sets[s].idx = (u32 *)q; q += e4;
            sets[s].slot = (u32 *)q; q += e4;
            sets[s].head = (u32 *)q; q += e4;
        }
        const int rb = CYCLIC ? bitlen_u64(n - 1) : bitlen_u64(n);
        if (2 * rb > 64) return BWTS_E_RANGE;
        const int round_key_bits = 2 * rb > 1 ? 2 * rb : 1;
        int nxt = 0;
        u32 *tpos = nullptr, *trank = nullptr;
        u64 *dir = nullptr;
        u32 *pdir = nullptr;
        int dlog = 0, psh = 0;
        const u64 a0 = a;
        if (sparse) {
            SpanGuard g(ctx, BWTS_K_RERANK, a, 24 * a);
            tpos = (u32 *)(base + 2 * e8 + 7 * e4);          // the last two arrays of the aux block
            trank = (u32 *)(base + 2 * e8 + 8 * e4);
            // directory over the sorted keys' top bits for the rank searches of keybuild_sparse_kernel
            const bool no_dir = [ctx] { const char *e = bwts_knob(ctx, "BWTS_K0DIR"); return e && atoi(e) == 0; }();
            const int kb = al.key_bits;
            dlog = kb < K0_DIR_LOG2_MAX ? kb : K0_DIR_LOG2_MAX;
            if (dlog > bitlen_u64(n)) dlog = bitlen_u64(n);
            if (!no_dir && dlog >= 8) {
                dir = (u64 *)(base + 2 * e8 + 9 * e4);
                k0_directory_kernel<<<dim3((unsigned)(((1ull << dlog) + 1 + 255) / 256)), dim3(256), 0, ctx->stream>>>(K0, n, kb, dlog, dir);
            }
            tied_map_keys_kernel<<<dim3((unsigned)((a + 255) / 256)), dim3(256), 0, ctx->stream>>>(cur.idx, a, akeys[0]);
            HIPC(hipMemcpyAsync(scratch, cur.head, a * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
            SortPlan mp;
            mp.keys[0] = akeys[0]; mp.keys[1] = akeys[1];
            mp.vals[0] = scratch; mp.vals[1] = trank;
            mp.tile_hist = sp.tile_hist; mp.scan_temp = sp.scan_temp;
            int mr = 0;
            BWTS_TRY(radix_sort_pairs(ctx, mp, a, bitlen_u64(n - 1) > 0 ? bitlen_u64(n - 1) : 1, &mr));
            tied_map_finish_kernel<<<dim3((unsigned)((a + 255) / 256)), dim3(256), 0, ctx->stream>>>(akeys[mr], a, tpos);
            if (mr == 0) HIPC(hipMemcpyAsync(trank, scratch, a * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
            if (dir) {          // same switch as the key directory: a directory over the map's positions
                const int pb = bitlen_u64(n - 1);
                psh = pb > K0_DIR_LOG2_MAX ? pb - K0_DIR_LOG2_MAX : 0;
                const u64 buckets = ((n - 1) >> psh) + 1;
                pdir = (u32 *)(base + 2 * e8 + 9 * e4 + align_up(((size_t)1 << K0_DIR_LOG2_MAX) * 8 + 8, 256));
                tpos_directory_kernel<<<dim3((unsigned)((buckets + 1 + 255) / 256)), dim3(256), 0, ctx->stream>>>(tpos, a, psh, buckets, pdir);
            }
            HIPC(hipGetLastError());
        } else {
            const bool plain_build = [ctx] { const char *e = bwts_knob(ctx, "BWTS_RANKBUILD"); return e && !strcmp(e, "plain"); }();
            BWTS_TRY(ensure_rank(ctx, sp, n));
            if (rank_early) {
                // built before the tied list (see above)
            } else if (flag_heads && n >= (1ull << 22) && !plain_build) {
                // BWTS_RANKBUILD=partition: the first binned form.  Pairs go through the sorted keys' buffer (not needed without the sparse map), counters through the tile table
                SpanGuard g(ctx, BWTS_K_RERANK, n, 28 * n);
                int wlog = bitlen_u64(n - 1) - 12;
                if (wlog < 18) wlog = 18;
                const u32 nw = (u32)((n + (1ull << wlog) - 1) >> wlog);
                u32 *win_fill = sp.tile_hist;
                HIPC(hipMemsetAsync(win_fill, 0, (size_t)nw * RB_FILL_STRIDE * sizeof(u32), ctx->stream));
                BWTS_TRY(ensure_dyn_lds(ctx, (const void *)rank_partition_kernel, rank_partition_lds_bytes(RB_MAX_WINDOWS)));
                rank_partition_kernel<<<dim3((unsigned)((n + RB_CHUNK - 1) / RB_CHUNK)), dim3(RB_THREADS), rank_partition_lds_bytes(nw), ctx->stream>>>(
                    SA, n, flag_heads, flag_pre, wlog, nw, win_fill, K0);
                rank_scatter_pairs_kernel<<<dim3(nw * RB_BLOCKS_PER_WINDOW), dim3(256), 0, ctx->stream>>>(K0, win_fill, wlog, sp.rank);
                HIPC(hipGetLastError());
            } else {
                BWTS_TRY(build_ranks(ctx, SA, n, cur, a, sp.rank));
            }
            rank_valid = true;
            // group-local rounds; SA is only rebuilt when someone reads it afterwards (suffix array requested, or the
            // gather form of the emission)
            const bool need_sa = !CYCLIC || !sp.carry_out;
            // chunks (chunk_rounds.h) unless BWTS_DENSE=tiles asks for the tile form (dense_rounds.h), the list is short or memory is
            const bool tiles_only = [ctx] { const char *e = bwts_knob(ctx, "BWTS_DENSE"); return e && !strcmp(e, "tiles"); }();
            bool handled = false;
            if (!tiles_only) BWTS_TRY((chunk_rounds<CYCLIC>(ctx, d_T, n, al, d_fstart, k, sp, cur, a, SA, need_sa, &rounds, &handled)));
            if (!handled) BWTS_TRY((dense_rounds<CYCLIC>(ctx, d_T, n, al, d_fstart, k, sp, cur, a, SA, need_sa, &rounds)));
            sp.ties_emitted = CYCLIC && sp.carry_out;
            *sa_out = SA;
            *rounds_out = rounds;
            return BWTS_OK;
        }

        // ---- few ties: sparse ranks, the list in SA order, a sort per round ----

        bool seg_skip_next = false;
        for (u64 h = (u64)al.hstep;; h <<= 1) {
            rounds++;
            {
                SpanGuard g(ctx, BWTS_K_KEYBUILD, a, 20 * a);
                const unsigned blocks = (unsigned)((a + 255) / 256);
                keybuild_sparse_kernel<CYCLIC><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(
                    cur.idx, cur.head, a, d_T, n, d_codes, al.bits, al.msym, al.pad_add, h, K0, rb, d_fstart, k, akeys[0],
                    al.varlen ? ctx->d_small + SM_VTAB : nullptr, al.key_bits, tpos, trank, a0, dir, dlog, pdir, psh);
                HIPC(hipGetLastError());
            }
            const u64 *AK;
            const u32 *AV;
            // (large groups dominating one round dominate the next one too: then the classification is skipped every other round)
            const bool seg_probe = !seg_skip_next;
            seg_skip_next = false;
            if (a > 4096 && seg_probe) {
                // groups of <= SEG_CAP in place, the rest through the radix sort (see seg_small_sort_kernel)
                char *sb = base + 2 * e8 + 9 * e4 + dir_bytes;
                u8 *big = (u8 *)sb;
                u64 *bk[2] = {(u64 *)(sb + e1), (u64 *)(sb + e1 + e8)};
                u32 *bv[2] = {(u32 *)(sb + e1 + 2 * e8), (u32 *)(sb + e1 + 2 * e8 + e4)};
                u32 *bpos = (u32 *)(sb + e1 + 2 * e8 + 2 * e4);
                u64 m_big = 0;
                {
                    SpanGuard g(ctx, BWTS_K_RERANK, a, 20 * a);
                    u64 *segcnt = ctx->d_small + SM_SEGCNT;
                    HIPC(hipMemsetAsync(segcnt, 0, 256 * sizeof(u64), ctx->stream));
                    const u64 waves = (a + SEG_OWN - 1) / SEG_OWN;
                    seg_small_sort_kernel<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, ctx->stream>>>(cur.head, akeys[0], cur.idx, a, big, segcnt);
                    HIPC(hipGetLastError());
                }
                BWTS_TRY(read_small(ctx, SM_SEGCNT, 256));
                for (int c = 0; c < 256; c++) m_big += ctx->h_small[SM_SEGCNT + c];
                if (m_big > a) return BWTS_E_INTERNAL;
                seg_skip_next = m_big * 10 > a * 9;
                if (m_big * 5 > a * 4) {
                    // larger groups hold nearly all of the list: sorting everything costs less than compacting them
                    // (by bytes moved the break-even is near 85 %)
                    SortPlan ap;
                    ap.keys[0] = akeys[0]; ap.keys[1] = akeys[1];
                    ap.vals[0] = cur.idx; ap.vals[1] = scratch;
                    ap.tile_hist = sp.tile_hist; ap.scan_temp = sp.scan_temp;
                    int r2 = 0;
                    BWTS_TRY(radix_sort_pairs(ctx, ap, a, round_key_bits, &r2));
                    AK = akeys[r2];
                    AV = r2 ? scratch : cur.idx;
                } else {
                    if (m_big) {
                        {
                            SpanGuard g(ctx, BWTS_K_RERANK, a, 20 * a);
                            SegFlagIn fin{big, cur.head};
                            SegBigOut fout{big, cur.head, akeys[0], cur.idx, a, rb, bk[0], bv[0], bpos, cnt + 3};
                            BWTS_TRY((device_scan<false, u64>(ctx, a, fin, fout, OpAdd(), (u64)0, sp.scan_temp)));
                        }
                        SortPlan bp;
                        bp.keys[0] = bk[0]; bp.keys[1] = bk[1];
                        bp.vals[0] = bv[0]; bp.vals[1] = bv[1];
                        bp.tile_hist = sp.tile_hist; bp.scan_temp = sp.scan_temp;
                        int rbig = 0;
                        const int big_bits = bitlen_u64(m_big / (SEG_CAP + 1)) + rb;       // ordinals < m_big / (SEG_CAP + 1)
                        BWTS_TRY(radix_sort_pairs(ctx, bp, m_big, big_bits < round_key_bits ? big_bits : round_key_bits, &rbig));
                        SpanGuard g(ctx, BWTS_K_RERANK, m_big, 28 * m_big);
                        seg_writeback_kernel<<<dim3((unsigned)((m_big + 255) / 256)), dim3(256), 0, ctx->stream>>>(bk[rbig], bv[rbig], bpos, m_big, cur.head, rb,
                                                                                                                 akeys[0], cur.idx);
                        HIPC(hipGetLastError());
                    }
                    AK = akeys[0];
                    AV = cur.idx;
                }
            } else {
                SortPlan ap;
                ap.keys[0] = akeys[0]; ap.keys[1] = akeys[1];
                ap.vals[0] = cur.idx; ap.vals[1] = scratch;
                ap.tile_hist = sp.tile_hist; ap.scan_temp = sp.scan_temp;
                int r2 = 0;
                BWTS_TRY(radix_sort_pairs(ctx, ap, a, round_key_bits, &r2));
                AK = akeys[r2];
                AV = r2 ? scratch : cur.idx;
            }

            HIPC(hipMemsetAsync(cnt, 0, 4 * sizeof(u64), ctx->stream));
            {
                SpanGuard g(ctx, BWTS_K_RERANK, a, 24 * a);
                GroupIn in{AK, cur.slot, a, rb};
                GroupOut out{cur.slot, AV, a, rb, AK, nullptr, tpos, trank, a0, SA,
                             sets[nxt].idx, sets[nxt].slot, sets[nxt].head, cnt + 0, cnt + 1};
                BWTS_TRY((device_scan<true, u64>(ctx, a, in, out, OpHeadCount(), (u64)0, sp.scan_temp)));
            }
            BWTS_TRY(read_small(ctx, SM_COUNTERS, 4));
            const u64 a_new = ctx->h_small[CNT_ACTIVE];
            const u64 splits = ctx->h_small[CNT_SPLITS];
            cur = sets[nxt];
            nxt ^= 1;
            a = a_new;
            if (CYCLIC && rounds - 1 < BWTS_MAX_ROUND_STATS) ctx->tm.round_active[rounds - 1] = a;
            if (a == 0) break;
            if (CYCLIC && splits == 0) break;               // partition stable under doubling: equal infinite words
            if (!CYCLIC && h >= n) return BWTS_E_INTERNAL;  // suffixes are distinct; cannot happen
            if (rounds > 80) return BWTS_E_INTERNAL;
        }
    }
    if (want_ranks && !rank_valid) { BWTS_TRY(ensure_rank(ctx, sp, n)); BWTS_TRY(build_ranks(ctx, SA, n, a ? cur : none, a, sp.rank)); }
    *sa_out = SA;
    *rounds_out = rounds;
    return BWTS_OK;
}

static_assert(KB_TILE == SCAN_TILE, "keybuild0's tile minima feed the scan's final sweep");

// keys of positions [pos0, pos0 + count) into keys0 (index q - pos0), tile minima into tile_min[0 ..)
static int launch_keybuild0_seg(bwts_ctx *ctx, const u8 *d_T, u64 n, const Alphabet &al, u64 *keys0, u64 *tile_min, bool split, u64 pos0, u64 count,
                                u8 *wprev = nullptr /* wide keys only: T[q - 1] of every position beside its key */)
{
    SpanGuard g(ctx, BWTS_K_KEYBUILD, count, count + (split ? 5 : 8) * count);
    const u64 blocks = (count + KB_TILE - 1) / KB_TILE;
    KeyStore ks = key_store_of(keys0, count, split, al.key_bits);
    ks.wprev = split ? nullptr : wprev;
    if (al.varlen) {
        keybuild0v_kernel<<<dim3((unsigned)blocks), dim3(KB_THREADS), 0, ctx->stream>>>(d_T, n, ctx->d_small + SM_VTAB, al.key_bits,
                                                                                        ks, tile_min, pos0, pos0 + count);
        HIPC(hipGetLastError());
        return BWTS_OK;
    }
    keybuild0_kernel<<<dim3((unsigned)blocks), dim3(KB_THREADS), 0, ctx->stream>>>(
        d_T, n, (const u8 *)(ctx->d_small + SM_CODES), al.bits, al.msym, al.pad_add, ks, tile_min, pos0, pos0 + count);
    HIPC(hipGetLastError());
    return BWTS_OK;
}
static int launch_keybuild0(bwts_ctx *ctx, const u8 *d_T, u64 n, const Alphabet &al, SortSpace &sp, u64 *tile_min, bool split)
{
    return launch_keybuild0_seg(ctx, d_T, n, al, sp.keys[0], tile_min, split, 0, n);
}

// ------------------------------------------------------------------------------------
// Lyndon factors, general path: strict prefix minima of the suffix ranks (mk_bwts_sa.c:126-129)
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

// suffix sort on the padded alphabet; SA in *d_sa, ISA in sp.rank when want_ranks
static int suffix_sort_in(bwts_ctx *ctx, const u8 *d_T, u64 n, SortSpace &sp, bool want_ranks, u32 **d_sa, u32 *rounds)
{
    if (n > 0xffffffffull) return BWTS_E_RANGE;
    Alphabet al;
    BWTS_TRY(set_alphabet(ctx, true, n, &al));
    BWTS_TRY(launch_keybuild0(ctx, d_T, n, al, sp, nullptr, false));
    u64 active0 = 0;
    return doubling_sort<false>(ctx, d_T, n, al, nullptr, 0, sp, want_ranks, d_sa, rounds, &active0);
}

int suffix_sort_device(bwts_ctx *ctx, const u8 *d_T, u64 n, u32 **d_sa, u32 **d_rank, u32 *rounds)
{
    BWTS_TRY(read_histogram(ctx, d_T, n));
    SortSpace sp;
    BWTS_TRY(sort_space_alloc(ctx, n, &sp));
    BWTS_TRY(suffix_sort_in(ctx, d_T, n, sp, true, d_sa, rounds));
    *d_rank = sp.rank;
    return BWTS_OK;
}

static int lyndon_general(bwts_ctx *ctx, const u8 *d_T, u64 n, SortSpace &sp, u32 **d_fstart, u64 *k_out, u32 *rounds)
{
    u32 *sa = nullptr;
    BWTS_TRY(suffix_sort_in(ctx, d_T, n, sp, true, &sa, rounds));
    // the sort's key buffers are free again: flags and the compacted starts live there
    u8 *flag = (u8 *)sp.keys[0];
    u32 *starts_tmp = (u32 *)sp.keys[1];
    u64 *total = ctx->d_small + CNT_TOTAL;
    {
        SpanGuard g(ctx, BWTS_K_LYNDON, n, 8 * n);
        RankIn in{sp.rank};
        MinFlagOut out{sp.rank, flag};
        BWTS_TRY((device_scan<false, u32>(ctx, n, in, out, OpMin(), 0xffffffffu, sp.scan_temp)));
        FlagIn fin{flag};
        StartOut sout{flag, starts_tmp, n, total};
        BWTS_TRY((device_scan<false, u32>(ctx, n, fin, sout, OpAdd(), 0u, sp.scan_temp)));
    }
    BWTS_TRY(read_small(ctx, CNT_TOTAL, 1));
    const u64 k = ctx->h_small[CNT_TOTAL];
    if (k == 0 || k > n) return BWTS_E_INTERNAL;
    char *fl = nullptr;
    BWTS_TRY(aux_reserve_slot(ctx, 2, (size_t)k * 4, &fl));
    u32 *dst = (u32 *)fl;
    HIPC(hipMemcpyAsync(dst, starts_tmp, k * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
    *d_fstart = dst;
    *k_out = k;
    return BWTS_OK;
}

// ------------------------------------------------------------------------------------
// Lyndon factors, fast path
// ------------------------------------------------------------------------------------
// p starts a factor iff its suffix is smaller than every earlier suffix.  With K(p) the packed
// first msym symbols: K(p) > min_{q<p} K(q) rules p out; K(p) < that minimum (and no zero fill
// in K(p)) proves it; equality -- or fill -- is settled by an exact comparison with the latest
// factor start.  Prefix-min scan over the round-0 keys -> candidate list -> one workgroup.
#define LYN_CAND_CAP   65536ull
#define LYN_WORK_CAP   (64ull << 20)      // bytes compared inside the resolver workgroup before it gives up

struct KeyIn { KeyStore K; __device__ __forceinline__ u64 operator()(u64 i) const { return ks_load(K, i); } };
struct CandOut {
    KeyStore K; u64 n; int msym; u64 *cand; u64 cap; u64 *counter;
    u64 pos0 = 0;          // element j of this launch is position pos0 + j (wide path: one segment at a time)
    __device__ __forceinline__ void operator()(u64 j, u64 min_before) const
    {
        const u64 ki = ks_load(K, j);
        const u64 i = pos0 + j;
        const bool c = i == 0 || ki <= min_before;
        const u64 m = __ballot(c);
        if (m == 0) return;
        const int leader = __ffsll((unsigned long long)m) - 1;
        unsigned long long base = 0;
        if (lane_id() == leader) base = atomicAdd((unsigned long long *)counter, (unsigned long long)__popcll(m));
        base = shfl_t((u64)base, leader);
        if (c) {
            const u64 at = base + (u64)__popcll(m & lanemask_lt());
            const bool definite = i == 0 || (ki < min_before && i + (u64)msym <= n);     // msym: symbols a key may reach over
            if (at < cap) cand[at] = (i << 1) | (definite ? 1ull : 0ull);
        }
    }
};

struct TileMayHoldCandidate {
    const u64 *tile_min;
    __device__ __forceinline__ bool operator()(u64 t, u64 min_before) const { return tile_min[t] <= min_before; }
};

// one workgroup walks the position-sorted candidates; exact suffix comparison is block-wide.  A comparison
// that is still undecided after LYN_LOCAL_LCE bytes is handed to the host, which runs it on the whole chip
// (suffix_less_grid_kernel) and restarts the walk behind that candidate.
#define LYN_LOCAL_LCE  (64u << 10)
struct LynState {         // in d_small: resumable state of the resolver
    u64 next;             // next candidate to look at
    u64 cur;              // latest factor start
    u64 k;                // factor starts found so far
    u64 status;           // 0 = finished, 1 = long comparison wanted (candidate `next`), 2 = work cap exceeded
    u64 forced;           // 1: the host has decided candidate `next`: forced_less says whether it starts a factor
    u64 forced_less;
    u64 work;
};

template <typename POS>
__global__ __launch_bounds__(256) void lyndon_resolve_kernel(const u8 *__restrict__ T, u64 n, const u64 *__restrict__ cand, u64 cnt,
                                                             POS *__restrict__ fstart, LynState *__restrict__ st, u64 work_cap)
{
    __shared__ int s_mis[4];      // per wave: first mismatching lane of the chunk, or -1
    __shared__ int s_less[4];     // per wave: candidate byte < current-start byte at that lane
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    u64 cur = st->cur, k = st->k, work = st->work;
    const u64 first = st->next;
    const bool forced = st->forced != 0, forced_less = st->forced_less != 0;
    __syncthreads();
    for (u64 c = first; c < cnt; c++) {
        const u64 e = cand[c];
        const u64 p = e >> 1;
        bool is_start;
        if (p == 0 || (e & 1)) {
            is_start = true;
        } else if (forced && c == first) {
            is_start = forced_less;
        } else {
            is_start = false;
            for (u64 off = 0;; off += 256) {
                const u64 i = off + tid;
                const int a = p + i < n ? (int)T[p + i] : -1;       // the shorter suffix is the smaller one
                const int b = cur + i < n ? (int)T[cur + i] : -1;
                const u64 mm = __ballot(a != b);
                if (lane == 0) s_mis[w] = mm ? __ffsll((unsigned long long)mm) - 1 : -1;
                if (mm && lane == __ffsll((unsigned long long)mm) - 1) s_less[w] = a < b;
                __syncthreads();
                int found = -1;
#pragma unroll
                for (int ww = 3; ww >= 0; ww--) if (s_mis[ww] >= 0) found = ww;
                const int less = found >= 0 ? s_less[found] : 0;
                __syncthreads();
                work += 256;
                if (found >= 0) { is_start = less != 0; break; }
                if (off + 256 >= LYN_LOCAL_LCE || work > work_cap) {
                    if (tid == 0) { st->next = c; st->cur = cur; st->k = k; st->work = work; st->forced = 0; st->status = work > work_cap ? 2 : 1; }
                    return;
                }
            }
        }
        if (is_start) {
            if (tid == 0) fstart[k] = (POS)p;
            k++;
            cur = p;
        }
    }
    if (tid == 0) { st->next = cnt; st->cur = cur; st->k = k; st->work = work; st->forced = 0; st->status = 0; }
}

// first position where T[p..] and T[q..] differ (q < p), over the whole chip: result[0] = min mismatch offset
__global__ __launch_bounds__(256) void suffix_mismatch_grid_kernel(const u8 *__restrict__ T, u64 n, u64 p, u64 q,
                                                                   unsigned long long *__restrict__ result)
{
    const u64 len = n - p;                  // p's suffix is the shorter one; offset len is a mismatch by definition
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < len; i += (u64)gridDim.x * 256) {
        if (i >= __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;     // someone found an earlier one
        if (T[p + i] != T[q + i]) { atomicMin(result, (unsigned long long)i); break; }
    }
}

// returns BWTS_OK with *done = false when the input needs the general path
static int lyndon_fast(bwts_ctx *ctx, const u8 *d_T, u64 n, const Alphabet &al, SortSpace &sp, const u64 *tile_min, u64 *cand[2],
                       u32 *cvals[2], u32 *fstart, u64 *k_out, bool *done)
{
    *done = false;
    u64 *cnt = ctx->d_small + SM_COUNTERS;
    HIPC(hipMemsetAsync(cnt + 4, 0, 4 * sizeof(u64), ctx->stream));
    {
        // keybuild0 left every tile's smallest key in tile_min; after the exclusive min-scan of those, a tile can
        // hold a candidate only if its own minimum does not exceed the minimum of everything before it
        SpanGuard g(ctx, BWTS_K_LYNDON, n, 0);
        const u64 tiles = scan_tiles(n);
        HIPC(hipMemcpyAsync(sp.scan_temp, tile_min, tiles * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
        BWTS_TRY((device_scan_partials<u64, OpMin>(ctx, tiles, OpMin(), ~0ull, sp.scan_temp)));
        const KeyStore ks = key_store_of(sp.keys[0], n, sp.split_keys, al.key_bits);
        KeyIn in{ks};
        CandOut out{ks, n, al.varlen ? 64 : al.msym, cand[0], LYN_CAND_CAP, ctx->d_small + CNT_CAND};
        TileMayHoldCandidate filter{tile_min};
        BWTS_TRY((device_scan_final<false, u64>(ctx, n, in, out, OpMin(), ~0ull, sp.scan_temp, filter)));
    }
    BWTS_TRY(read_small(ctx, CNT_CAND, 1));
    const u64 cnt_c = ctx->h_small[CNT_CAND];
    if (cnt_c == 0) return BWTS_E_INTERNAL;
    if (cnt_c > LYN_CAND_CAP) return BWTS_OK;
    // candidates arrive in arbitrary order: sort by position
    SortPlan cp;
    cp.keys[0] = cand[0]; cp.keys[1] = cand[1];
    cp.vals[0] = cvals[0]; cp.vals[1] = cvals[1];
    cp.tile_hist = sp.tile_hist; cp.scan_temp = sp.scan_temp;
    int res = 0;
    BWTS_TRY(radix_sort_pairs(ctx, cp, cnt_c, bitlen_u64(n) + 1, &res));
    // resumable resolver: long comparisons are decided by a grid-wide kernel between two of its launches
    LynState *d_st = (LynState *)(ctx->d_small + CNT_LYN_K);
    LynState *h_st = (LynState *)(ctx->h_small + CNT_LYN_K);
    unsigned long long *d_mis = (unsigned long long *)(ctx->d_small + CNT_LYN_K + 8);
    HIPC(hipMemsetAsync(d_st, 0, sizeof(LynState), ctx->stream));
    for (int iter = 0;; iter++) {
        {
            SpanGuard g(ctx, BWTS_K_LYNDON, cnt_c, 0);
            lyndon_resolve_kernel<u32><<<dim3(1), dim3(256), 0, ctx->stream>>>(d_T, n, cand[res], cnt_c, fstart, d_st, LYN_WORK_CAP);
            HIPC(hipGetLastError());
        }
        BWTS_TRY(read_small(ctx, CNT_LYN_K, 8));
        if (h_st->status == 0) break;
        if (h_st->status == 2 || iter > 4096) return BWTS_OK;       // too much sequential work: general path
        // status 1: compare suffix(p) with suffix(cur) on the whole chip
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
        if (at >= len) less = true;                                 // suffix(p) is a proper prefix of suffix(cur): the shorter is smaller
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
    *k_out = h_st->k;
    if (*k_out == 0) return BWTS_E_INTERNAL;
    *done = true;
    return BWTS_OK;
}

static int lyndon_mode(const bwts_ctx *ctx)
{
    const char *env = bwts_knob(ctx, "BWTS_LYNDON");     // auto (default) | fast | general
    if (env && !strcmp(env, "general")) return 2;
    if (env && !strcmp(env, "fast")) return 1;
    return 0;
}

// Finds the factors and leaves the cyclic round-0 keys in sp.keys[0] / identity in sp.vals[0].
static int factors_and_keys(bwts_ctx *ctx, const u8 *d_T, u64 n, SortSpace &sp, Alphabet *al, u32 **d_fstart, u64 *k_out,
                            u32 *lyndon_rounds, bool hist_ready = false)
{
    u64 *cand[2];
    u32 *cvals[2];
    for (int i = 0; i < 2; i++) {
        cand[i] = arena_array<u64>(ctx, LYN_CAND_CAP);
        cvals[i] = arena_array<u32>(ctx, LYN_CAND_CAP);
        if (!cand[i] || !cvals[i]) return BWTS_E_NOMEM;
    }
    u32 *fast_starts = arena_array<u32>(ctx, LYN_CAND_CAP);
    if (!fast_starts) return BWTS_E_NOMEM;

    if (!hist_ready) BWTS_TRY(read_histogram(ctx, d_T, n));
    const int mode = lyndon_mode(ctx);
    bool done = false;
    *lyndon_rounds = 0;
    if (mode != 2) {
        u64 *tile_min = arena_array<u64>(ctx, scan_tiles(n) + 1);
        if (!tile_min) return BWTS_E_NOMEM;
        SampleScratch ss{d_T, {sp.keys[0], sp.keys[1]}, {sp.vals[0], sp.vals[1]}, sp.tile_hist, sp.scan_temp};
        BWTS_TRY(set_alphabet(ctx, false, n, al, &ss));
        sp.split_keys = sp.want_split && radix_packed_applicable(ctx, n, al->key_bits);
        STAGE("histogram + alphabet");
        BWTS_TRY(launch_keybuild0(ctx, d_T, n, *al, sp, tile_min, sp.split_keys));
        STAGE("keybuild0");
        BWTS_TRY(lyndon_fast(ctx, d_T, n, *al, sp, tile_min, cand, cvals, fast_starts, k_out, &done));
        STAGE("lyndon_fast");
        if (done) *d_fstart = fast_starts;
        else if (mode == 1) return BWTS_E_INTERNAL;
    }
    if (!done) {
        BWTS_TRY(lyndon_general(ctx, d_T, n, sp, d_fstart, k_out, lyndon_rounds));
        SampleScratch ss{d_T, {sp.keys[0], sp.keys[1]}, {sp.vals[0], sp.vals[1]}, sp.tile_hist, sp.scan_temp};
        BWTS_TRY(set_alphabet(ctx, false, n, al, &ss));
        sp.split_keys = sp.want_split && radix_packed_applicable(ctx, n, al->key_bits);
        BWTS_TRY(launch_keybuild0(ctx, d_T, n, *al, sp, nullptr, sp.split_keys));
    }
    const KeyStore ks = key_store_of(sp.keys[0], n, sp.split_keys, al->key_bits);
    if (sp.split_keys) {
        carried_head_fix_kernel<<<dim3((unsigned)((*k_out + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_T, n, *d_fstart, *k_out, ks);
        HIPC(hipGetLastError());
        STAGE("carried_head_fix");
    }
    // wrap the keys of positions near their factor's end
    if (al->varlen) {
        SpanGuard g(ctx, BWTS_K_KEYBUILD, *k_out * 64, 0);
        const u64 threads = *k_out * 64;
        cyclic_patch_vl_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream>>>(
            d_T, n, ctx->d_small + SM_VTAB, al->key_bits, *d_fstart, *k_out, ks);
        HIPC(hipGetLastError());
    } else if (al->msym > 1) {
        SpanGuard g(ctx, BWTS_K_KEYBUILD, *k_out * (u64)(al->msym - 1), 0);
        const u64 threads = *k_out * (u64)(al->msym - 1);
        cyclic_patch_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream>>>(
            d_T, n, (const u8 *)(ctx->d_small + SM_CODES), al->bits, al->msym, *d_fstart, *k_out, ks);
        HIPC(hipGetLastError());
    }
    return BWTS_OK;
}

int lyndon_factors_device(bwts_ctx *ctx, const u8 *d_T, u64 n, u32 **d_fstart, u64 *k_out, u32 *rounds)
{
    SortSpace sp;
    BWTS_TRY(sort_space_alloc(ctx, n, &sp));
    Alphabet al;
    return factors_and_keys(ctx, d_T, n, sp, &al, d_fstart, k_out, rounds);
}

// ------------------------------------------------------------------------------------
// emission (mk_bwts_sa.c:172-188): P[p] = T[cprev(p)], bwts[r] = P[sa[r]]
// ------------------------------------------------------------------------------------
// P[i] = T[i-1] (P[0] = T[n-1]); 16 bytes per thread: one 16-byte load, the byte in front of it, one 16-byte store
__global__ __launch_bounds__(256) void prevsym_kernel(const u8 *__restrict__ T, u64 n, u8 *__restrict__ P)
{
    const bool vec_ok = (((uintptr_t)T | (uintptr_t)P) & 15) == 0;
    for (u64 c = (u64)blockIdx.x * 256 + threadIdx.x; c * 16 < n; c += (u64)gridDim.x * 256) {
        const u64 i = c * 16;
        if (vec_ok && i + 16 <= n) {
            const uint4 v = *(const uint4 *)(T + i);
            const u32 before = i ? (u32)T[i - 1] : (u32)T[n - 1];
            uint4 o;
            o.x = (v.x << 8) | before;
            o.y = (v.y << 8) | (v.x >> 24);
            o.z = (v.z << 8) | (v.y >> 24);
            o.w = (v.w << 8) | (v.z >> 24);
            *(uint4 *)(P + i) = o;
        } else {
            for (u64 j = i; j < n && j < i + 16; j++) P[j] = j ? T[j - 1] : T[n - 1];
        }
    }
}
__global__ __launch_bounds__(256) void prevsym_fix_kernel(const u8 *__restrict__ T, u64 n, const u32 *__restrict__ fstart, u64 k,
                                                          u8 *__restrict__ P)
{
    const u64 f = (u64)blockIdx.x * 256 + threadIdx.x;
    if (f < k) P[fstart[f]] = T[factor_end(fstart, k, n, f) - 1];
}
__global__ __launch_bounds__(256) void carried_head_fix_kernel(const u8 *__restrict__ T, u64 n, const u32 *__restrict__ fstart, u64 k, KeyStore ks)
{
    const u64 f = (u64)blockIdx.x * 256 + threadIdx.x;
    if (f >= k) return;
    const u64 p = fstart[f];
    const u8 last = T[factor_end(fstart, k, n, f) - 1];
    if (ks.c16) ks.c[2 * p + 1] = last; else ks.c[p] = last;
}
__global__ __launch_bounds__(256) void emit_kernel(const u32 *__restrict__ SA, const u8 *__restrict__ P, u64 n, u8 *__restrict__ out)
{
    // 8 slots per thread: two 16-byte index loads, eight independent gathers in flight, one packed 8-byte store
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 i = q * 8;
    if (i + 8 <= n && ((uintptr_t)out & 7) == 0) {
        const uint4 s0 = *(const uint4 *)(SA + i), s1 = *(const uint4 *)(SA + i + 4);
        const u32 b0 = P[s0.x], b1 = P[s0.y], b2 = P[s0.z], b3 = P[s0.w], b4 = P[s1.x], b5 = P[s1.y], b6 = P[s1.z], b7 = P[s1.w];
        const u32 lo = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24), hi = b4 | (b5 << 8) | (b6 << 16) | (b7 << 24);
        *(uint2 *)(out + i) = make_uint2(lo, hi);
    } else {
        for (u64 j = i; j < n && j < i + 8; j++) out[j] = P[SA[j]];
    }
}

// bytes of elements that the later rounds moved: out[slot] = P[sa[slot]]
// P == nullptr (split keys: no previous-symbol array was built): T[cprev(p)] is looked up through the factor list
__global__ __launch_bounds__(256) void patch_ties_kernel(const u32 *__restrict__ slots, u64 a, const u32 *__restrict__ SA,
                                                         const u8 *__restrict__ P, const u8 *__restrict__ T, u64 n,
                                                         const u32 *__restrict__ fstart, u64 k, u8 *__restrict__ out)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= a) return;
    const u32 r = slots[i];
    const u64 p = SA[r];
    if (P) { out[r] = P[p]; return; }
    const u64 f = factor_of(fstart, k, p);
    out[r] = fstart[f] == p ? T[factor_end(fstart, k, n, f) - 1] : T[p - 1];
}

#include "wide_path.h"

size_t forward_arena_bytes(u64 n)
{
    // candidate buffers + sort space + the round-0 flag words (packed passes; a sort on wide keys keeps the previous-symbol array
    // and two carried-byte buffers in a side block instead, like the general path's factor list and the dense rank array)
    return 8 * align_up(LYN_CAND_CAP * 8, 256) + sort_space_bytes(n) + 2 * align_up(n / 4 + 64, 256) + align_up(n / 256 + 64, 256) + (1 << 16);
}

// is the input one byte value repeated?  Eight probes, then -- only if they agree -- the byte histogram.
int constant_input_probe(bwts_ctx *ctx, const u8 *d_in, u64 n, bool *constant)
{
    *constant = false;
    u8 probe[8];
    for (int i = 0; i < 8; i++) HIPC(hipMemcpyAsync(&probe[i], d_in + (n - 1) / 7 * (u64)i, 1, hipMemcpyDefault, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    for (int i = 1; i < 8; i++) if (probe[i] != probe[0]) return BWTS_OK;
    BWTS_TRY(byte_histogram_device(ctx, d_in, n, ctx->d_small));
    BWTS_TRY(read_small(ctx, 0, 256));
    int present = 0;
    for (int c = 0; c < 256; c++) present += ctx->h_small[c] ? 1 : 0;
    *constant = present == 1;
    return BWTS_OK;
}

int forward_device_impl(bwts_ctx *ctx, const u8 *d_in, u64 n, u8 *d_out)
{
    // beyond 32-bit indices: the blocked path with 64-bit positions and ranks (wide_path.h).  BWTS_FORCE_WIDE=1 sends every
    // input there, falling back when the wide form cannot take it; =2 does not fall back (tests)
    const int force_wide = [ctx] { const char *e = bwts_knob(ctx, "BWTS_FORCE_WIDE"); return e ? atoi(e) : 0; }();
    if (n > 0x100000000ull) {
        // one byte value only, beyond 2^32 (every position is a factor and a candidate: the wide path's factor search would refuse it
        // with BWTS_E_RANGE): the identity, as below -- eight probes first, the histogram only if they agree
        bool constant = false;
        BWTS_TRY(constant_input_probe(ctx, d_in, n, &constant));
        if (constant) {
            HIPC(hipMemcpyAsync(d_out, d_in, n, hipMemcpyDefault, ctx->stream));
            ctx->tm.factors = n; ctx->tm.rounds = 1; ctx->tm.lyndon_rounds = 0; ctx->tm.active_after_round0 = 0;
            ctx->tm.key_symbols = 1; ctx->tm.key_bits = 1;
            return BWTS_OK;
        }
    }
    if (n > 0x100000000ull || force_wide) {
        const int rc = forward_wide_impl(ctx, d_in, n, d_out);
        if (n > 0x100000000ull || force_wide == 2 || rc != BWTS_E_RANGE) return rc;
    }
    // one byte value only: n factors of one symbol, every rotation equal -- the transform is the identity (mk_bwts_sa.c:172-188 emits
    // each factor's own last byte).  Taken before anything is allocated: at n = 2^32 this is the input on which every position stays
    // tied, which the tied-list buffers (32-bit slots) cannot hold.
    BWTS_TRY(read_histogram(ctx, d_in, n));
    {
        int present = 0;
        for (int c = 0; c < 256; c++) present += ctx->h_small[SM_HIST + c] ? 1 : 0;
        if (present == 1) {
            HIPC(hipMemcpyAsync(d_out, d_in, n, hipMemcpyDeviceToDevice, ctx->stream));
            ctx->tm.factors = n; ctx->tm.rounds = 1; ctx->tm.lyndon_rounds = 0; ctx->tm.active_after_round0 = 0;
            ctx->tm.key_symbols = 1; ctx->tm.key_bits = 1;
            return BWTS_OK;
        }
    }
    BWTS_TRY(arena_reserve(ctx, forward_arena_bytes(n)));
    SortSpace sp;
    BWTS_TRY(sort_space_alloc(ctx, n, &sp));

    // 1. Lyndon factors + round-0 keys
    Alphabet al;
    u32 *d_fstart = nullptr;
    u64 k = 0;
    u32 lrounds = 0;
    const char *emit_env = bwts_knob(ctx, "BWTS_EMIT");      // carry (default) | gather
    const bool carry = radix_supports_sym(ctx) && !(emit_env && !strcmp(emit_env, "gather"));
    sp.want_split = carry;                           // the byte stream rides round 0 => the packed passes may take split keys
    BWTS_TRY(factors_and_keys(ctx, d_in, n, sp, &al, &d_fstart, &k, &lrounds, true));
    ctx->tm.factors = k;
    ctx->tm.lyndon_rounds = lrounds;
    ctx->tm.key_symbols = (u32)al.msym;
    ctx->tm.key_bits = (u32)al.key_bits;

    // P[p] = T[cprev(p)] (mk_bwts_sa.c:172-188): a factor's head takes the factor's last byte.  With split keys the
    // byte already travels in the keys' c stream and no array is built.
    u8 *P = nullptr;
    char *pc = nullptr;                 // side block of a sort on wide keys: P, then the two carried-byte buffers
    const size_t n1 = align_up((size_t)n, 256);
    if (!sp.split_keys) {
        BWTS_TRY(aux_reserve_slot(ctx, 3, carry ? 3 * n1 : n1, &pc));
        P = (u8 *)pc;
        SpanGuard g(ctx, BWTS_K_OTHER, n, 2 * n);
        u64 blocks = (n / 16 + 255) / 256 + 1; if (blocks > 8192) blocks = 8192;
        prevsym_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(d_in, n, P);
        prevsym_fix_kernel<<<dim3((unsigned)((k + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_in, n, d_fstart, k, P);
        HIPC(hipGetLastError());
    }
    if (carry) {
        sp.carry_src = P;
        if (sp.split_keys) {
            // the byte travels in the keys' c stream: these two only hold the round-0 flag words (2 x n/8 and n/8 bytes)
            sp.carry_buf[0] = (u8 *)arena_alloc(ctx, (size_t)n / 4 + 64);
            sp.carry_buf[1] = (u8 *)arena_alloc(ctx, (size_t)n / 8 + 64);
        } else {
            sp.carry_buf[0] = (u8 *)pc + n1;
            sp.carry_buf[1] = (u8 *)pc + 2 * n1;
        }
        sp.carry_out = d_out;
        if (!sp.carry_buf[0] || !sp.carry_buf[1]) return BWTS_E_NOMEM;
    }

    // 2. cyclic sort
    u32 *SA = nullptr;
    u32 rounds = 0;
    u64 active0 = 0;
    BWTS_TRY((doubling_sort<true>(ctx, d_in, n, al, d_fstart, k, sp, false, &SA, &rounds, &active0)));
    ctx->tm.rounds = rounds;
    ctx->tm.active_after_round0 = active0;

    // 3. emission: either it rode on the sort (only the tied elements are patched), or a gather
    if (carry) {
        if (active0 && !sp.ties_emitted) {
            SpanGuard g(ctx, BWTS_K_EMIT, active0, 6 * active0);
            patch_ties_kernel<<<dim3((unsigned)((active0 + 255) / 256)), dim3(256), 0, ctx->stream>>>(sp.tie_slots, active0, SA, P, d_in, n, d_fstart, k, d_out);
        }
    } else {
        SpanGuard g(ctx, BWTS_K_EMIT, n, 6 * n);
        const u64 octs = (n + 7) / 8;
        emit_kernel<<<dim3((unsigned)((octs + 255) / 256)), dim3(256), 0, ctx->stream>>>(SA, P, n, d_out);
    }
    HIPC(hipGetLastError());
    return BWTS_OK;
}
