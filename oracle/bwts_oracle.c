/*
 * bwts_oracle.c -- CPU oracle for the bijective BWT (BWTS) hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see bwts_oracle.h).  Plain C restatement of what the
 * reference computes; file:line citations are relative to /root/reference.
 */
#define _POSIX_C_SOURCE 200809L
#include "bwts_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* Suffix sorter (stands in for libdivsufsort's divsufsort(), mk_bwts_sa.c:48) */
/* ------------------------------------------------------------------------- */
/* Induced sorting (SA-IS, Nong/Zhang/Chan) over a text that carries a virtual
 * unique smallest terminator: level 0 reads the caller's bytes shifted by +1
 * with symbol 0 at index n; deeper levels are int32 strings that already end in
 * their own unique 0. */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define IDX int32_t
#define N(x) x##_32
#include "bwts_oracle_core.h"
#undef IDX
#undef N
#define IDX int64_t
#define N(x) x##_64
#include "bwts_oracle_core.h"
#undef IDX
#undef N

int oracle_suffix_array(const uint8_t *T, int32_t *SA, int64_t n)
{
    return suffix_array_32(T, SA, n);
}

/* 32-bit indices (the reference's) below 2^31 - 1 bytes, 64-bit ones above */
int oracle_forward_timed(const uint8_t *T, int64_t n, uint8_t *out, double phase_s[4])
{
    if (n < 0x7fffffffLL) return forward_timed_32(T, n, out, phase_s);
    return forward_timed_64(T, n, out, phase_s);
}

int oracle_forward64(const uint8_t *T, int64_t n, uint8_t *out)
{
    return forward_timed_64(T, n, out, NULL);
}

int oracle_forward(const uint8_t *T, int64_t n, uint8_t *out)
{
    return oracle_forward_timed(T, n, out, NULL);
}

/* ------------------------------------------------------------------------- */
/* Forward transform from the definition                                      */
/* ------------------------------------------------------------------------- */

int64_t oracle_lyndon_starts(const uint8_t *T, int64_t n, int64_t *starts, int64_t cap)
{
    int64_t i = 0, cnt = 0;
    while (i < n) {
        int64_t j = i + 1, k = i;
        while (j < n && T[k] <= T[j]) {
            if (T[k] < T[j]) k = i; else k++;
            j++;
        }
        while (i <= k) {
            if (cnt < cap) starts[cnt] = i;
            cnt++;
            i += j - k;
        }
    }
    return cnt;
}

static const uint8_t *g_T;
static const int32_t *g_fs, *g_fe;    /* per position: its factor's [start,end) */

static int cmp_rot_omega(const void *pa, const void *pb)
{
    const int32_t p = *(const int32_t *)pa, q = *(const int32_t *)pb;
    const int32_t sp = g_fs[p], ep = g_fe[p], sq = g_fs[q], eq = g_fe[q];
    int64_t steps = (int64_t)(ep - sp) + (eq - sq);   /* Fine-Wilf bound */
    int32_t x = p, y = q;
    while (steps-- > 0) {
        if (g_T[x] != g_T[y]) return g_T[x] < g_T[y] ? -1 : 1;
        if (++x == ep) x = sp;
        if (++y == eq) y = sq;
    }
    return (p > q) - (p < q);   /* equal infinite words: any order gives the same bytes */
}

int oracle_forward_def(const uint8_t *T, int64_t n, uint8_t *out)
{
    int64_t k, f, i;
    int64_t *starts;
    int32_t *fs, *fe, *ord;
    if (n <= 0 || n >= 0x7fffffffLL) return -1;
    starts = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    fs = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    fe = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    ord = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!starts || !fs || !fe || !ord) { free(starts); free(fs); free(fe); free(ord); return -1; }
    k = oracle_lyndon_starts(T, n, starts, n);
    for (f = 0; f < k; f++) {
        const int64_t s = starts[f], e = f + 1 < k ? starts[f + 1] : n;
        for (i = s; i < e; i++) { fs[i] = (int32_t)s; fe[i] = (int32_t)e; }
    }
    for (i = 0; i < n; i++) ord[i] = (int32_t)i;
    g_T = T; g_fs = fs; g_fe = fe;
    qsort(ord, (size_t)n, sizeof(int32_t), cmp_rot_omega);
    for (i = 0; i < n; i++) {
        const int32_t p = ord[i];
        out[i] = T[p > fs[p] ? p - 1 : fe[p] - 1];
    }
    free(starts); free(fs); free(fe); free(ord);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Inverse transform, reference pipeline (unbwts.c:31-86)                     */
/* ------------------------------------------------------------------------- */

int oracle_inverse(const uint8_t *B, int64_t n, uint8_t *out)
{
    int64_t cnt[256], i, sum = 0, scan = 0, done = 0, wr;
    int32_t *lf;
    if (n <= 0 || n >= 0x7fffffffLL) return -1;
    lf = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!lf) return -1;
    memset(cnt, 0, sizeof cnt);
    for (i = 0; i < n; i++) cnt[B[i]]++;                  /* :34-36 */
    for (i = 0; i < 256; i++) {                           /* :38-43 */
        const int64_t c = cnt[i];
        cnt[i] = sum;
        sum += c;
    }
    for (i = 0; i < n; i++) lf[i] = (int32_t)cnt[B[i]]++; /* :50-52 */
    wr = n - 1;
    while (done < n) {                                    /* :66-86 */
        int64_t x;
        while (lf[scan] < 0) scan++;                      /* smallest unvisited index */
        x = scan;
        do {
            const int32_t nx = lf[x];
            out[wr--] = B[x];
            lf[x] = -1;
            x = nx;
            done++;
        } while (x != scan);
    }
    free(lf);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Synthetic inputs                                            */
/* ------------------------------------------------------------------------- */

static inline uint64_t splitmix_at(uint64_t seed, uint64_t k)   /* k-th output, k >= 1 */
{
    uint64_t z = seed + k * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static const char zipf_alpha[] =
    " etaoinshrdlcumwfgypbvkjxqz"
    "\n<>/=\"[]|'.,:;-_()&#0123456789"
    "ETAOINSHRDLCUMWFGYPBVKJXQZ"
    "!$%*+?@\\^`{}~";

/* kind 3, "text": the zipf stream with back-references -- the long repeats that real text has and an i.i.d. source lacks
 * (the reference's only named workload is enwik8, Makefile:35-38, which is not in this image).
 * Integer-only and seekable like the others.  For level l = 16 .. 4 the positions are cut into aligned windows of 2^l
 * bytes; window w >= 1 of level l is a COPY when the low bits of h = splitmix(seed + l * TEXT_LEVEL_SALT, w) are zero
 * (1 window in 8 for l = 4..7, 1 in 16 for l = 8..11, 1 in 32 for l = 12..16), and then its bytes equal the text at
 * src + (p mod 2^l), src = (h >> 8) mod ((w - 1) * 2^l + 1): any earlier, unaligned offset.  A position is redirected
 * through the largest copy window that holds it until it lies in none (every hop moves it towards 0); its byte is the
 * zipf byte of that final position.  About 60 % of the text is copied material, with repeats of 16 B .. 64 KiB, nested. */
#define TEXT_LEVEL_SALT 0xD1B54A32D192ED03ULL
static inline uint64_t text_resolve(uint64_t seed, uint64_t p)
{
    int l = 16;
    while (l >= 4) {
        const uint64_t w = p >> l;
        if (w) {
            const uint64_t h = splitmix_at(seed + (uint64_t)l * TEXT_LEVEL_SALT, w);
            const uint64_t mask = l < 8 ? 7u : l < 12 ? 15u : 31u;
            if ((h & mask) == 0) {
                const uint64_t span = ((w - 1) << l) + 1;
                p = (h >> 8) % span + (p & (((uint64_t)1 << l) - 1));
                l = 16;
                continue;
            }
        }
        l--;
    }
    return p;
}

void oracle_generate(int kind, uint64_t seed, uint64_t off, uint64_t len, uint8_t *dst)
{
    uint64_t i;
    if (kind == 0) {
        for (i = 0; i < len; i++) {
            const uint64_t o = off + i;
            dst[i] = (uint8_t)(splitmix_at(seed, o / 8 + 1) >> (8 * (o % 8)));
        }
    } else if (kind == 2) {
        for (i = 0; i < len; i++) {
            const uint64_t o = off + i;
            dst[i] = (uint8_t)"ACGT"[(splitmix_at(seed, o / 32 + 1) >> (2 * (o % 32))) & 3];
        }
    } else {
        uint64_t cum[96], W = 0;
        int k;
        for (k = 0; k < 96; k++) { W += (1u << 24) / (uint64_t)(k + 1); cum[k] = W; }
        for (i = 0; i < len; i++) {
            const uint64_t p = kind == 3 ? text_resolve(seed, off + i) : off + i;
            const uint64_t z = splitmix_at(seed, p + 1);
            const uint64_t u = ((z >> 32) * W) >> 32;
            int lo = 0, hi = 95;
            while (lo < hi) { int mid = (lo + hi) / 2; if (u < cum[mid]) hi = mid; else lo = mid + 1; }
            dst[i] = (uint8_t)zipf_alpha[lo];
        }
    }
}
