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

typedef struct {
    const uint8_t *b;   /* level-0 bytes, or NULL */
    const int32_t *w;   /* deeper-level symbols, or NULL */
    int64_t nb;         /* number of real bytes at level 0 */
} sym_src;

static inline int32_t sym_at(const sym_src *s, int64_t i)
{
    if (s->w) return s->w[i];
    return i < s->nb ? (int32_t)s->b[i] + 1 : 0;
}

#define TY_GET(i)  ((ty[(i) >> 3] >> ((i) & 7)) & 1)          /* 1 = S-type */
#define TY_SET(i)  (ty[(i) >> 3] |= (uint8_t)(1u << ((i) & 7)))
#define IS_LMS(i)  ((i) > 0 && TY_GET(i) && !TY_GET((i) - 1))

static void bucket_bounds(const sym_src *s, int64_t n, int32_t K, int32_t *bkt, int ends)
{
    int64_t i;
    int32_t sum = 0;
    for (i = 0; i < K; i++) bkt[i] = 0;
    for (i = 0; i < n; i++) bkt[sym_at(s, i)]++;
    for (i = 0; i < K; i++) {
        sum += bkt[i];
        bkt[i] = ends ? sum : sum - bkt[i];
    }
}

static void induce_l(const sym_src *s, const uint8_t *ty, int32_t *SA, int64_t n, int32_t K, int32_t *bkt)
{
    int64_t i;
    bucket_bounds(s, n, K, bkt, 0);
    for (i = 0; i < n; i++) {
        int64_t j = (int64_t)SA[i] - 1;
        if (j >= 0 && !TY_GET(j)) SA[bkt[sym_at(s, j)]++] = (int32_t)j;
    }
}

static void induce_s(const sym_src *s, const uint8_t *ty, int32_t *SA, int64_t n, int32_t K, int32_t *bkt)
{
    int64_t i;
    bucket_bounds(s, n, K, bkt, 1);
    for (i = n - 1; i >= 0; i--) {
        int64_t j = (int64_t)SA[i] - 1;
        if (j >= 0 && TY_GET(j)) SA[--bkt[sym_at(s, j)]] = (int32_t)j;
    }
}

static int sais_level(const sym_src *s, int32_t *SA, int64_t n, int32_t K)
{
    int64_t i, j, n1;
    int32_t name, prev;
    uint8_t *ty = (uint8_t *)calloc((size_t)(n / 8 + 1), 1);
    int32_t *bkt = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
    if (!ty || !bkt) { free(ty); free(bkt); return -1; }

    /* classify: last symbol (terminator) is S, the one before it is L */
    TY_SET(n - 1);
    for (i = n - 3; i >= 0; i--) {
        int32_t a = sym_at(s, i), b = sym_at(s, i + 1);
        if (a < b || (a == b && TY_GET(i + 1))) TY_SET(i);
    }

    /* stage 1: sort LMS substrings */
    bucket_bounds(s, n, K, bkt, 1);
    for (i = 0; i < n; i++) SA[i] = -1;
    for (i = 1; i < n; i++)
        if (IS_LMS(i)) SA[--bkt[sym_at(s, i)]] = (int32_t)i;
    induce_l(s, ty, SA, n, K, bkt);
    induce_s(s, ty, SA, n, K, bkt);

    n1 = 0;
    for (i = 0; i < n; i++)
        if (IS_LMS(SA[i])) SA[n1++] = SA[i];
    for (i = n1; i < n; i++) SA[i] = -1;

    name = 0; prev = -1;
    for (i = 0; i < n1; i++) {
        int32_t pos = SA[i];
        int diff = 0;
        int64_t d;
        for (d = 0; d < n; d++) {
            if (prev == -1 || sym_at(s, pos + d) != sym_at(s, prev + d) ||
                TY_GET(pos + d) != TY_GET(prev + d)) { diff = 1; break; }
            if (d > 0 && (IS_LMS(pos + d) || IS_LMS(prev + d))) break;
        }
        if (diff) { name++; prev = pos; }
        SA[n1 + pos / 2] = name - 1;
    }
    for (i = n - 1, j = n - 1; i >= n1; i--)
        if (SA[i] >= 0) SA[j--] = SA[i];

    /* stage 2: order the reduced string */
    {
        int32_t *SA1 = SA, *s1 = SA + n - n1;
        if (name < n1) {
            sym_src sub; sub.b = NULL; sub.w = s1; sub.nb = 0;
            if (sais_level(&sub, SA1, n1, name) != 0) { free(ty); free(bkt); return -1; }
        } else {
            for (i = 0; i < n1; i++) SA1[s1[i]] = (int32_t)i;
        }

        /* stage 3: induce the full order from the sorted LMS suffixes */
        bucket_bounds(s, n, K, bkt, 1);
        for (i = 1, j = 0; i < n; i++)
            if (IS_LMS(i)) s1[j++] = (int32_t)i;
        for (i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
        for (i = n1; i < n; i++) SA[i] = -1;
        for (i = n1 - 1; i >= 0; i--) {
            j = SA[i]; SA[i] = -1;
            SA[--bkt[sym_at(s, j)]] = (int32_t)j;
        }
    }
    induce_l(s, ty, SA, n, K, bkt);
    induce_s(s, ty, SA, n, K, bkt);

    free(ty); free(bkt);
    return 0;
}

int oracle_suffix_array(const uint8_t *T, int32_t *SA, int64_t n)
{
    int32_t *tmp;
    sym_src s;
    if (n <= 0) return 0;
    if (n == 1) { SA[0] = 0; return 0; }
    tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    if (!tmp) return -1;
    s.b = T; s.w = NULL; s.nb = n;
    if (sais_level(&s, tmp, n + 1, 257) != 0) { free(tmp); return -1; }
    /* tmp[0] is the terminator suffix; the rest is the SA with shorter-first ties */
    memcpy(SA, tmp + 1, sizeof(int32_t) * (size_t)n);
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Forward transform, reference pipeline                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t *T;
    int64_t n;
    int32_t *sa, *isa;
} fwd_state;

/* mk_bwts_sa.c:74-112 -- a factor's head moves from its suffix rank to its
 * rotation rank: it slides right past every later-in-text suffix that the
 * factor (repeated) is not smaller than. */
static int32_t settle_head(fwd_state *st, int32_t head, int32_t flen, int32_t rank)
{
    const int64_t n = st->n;
    while (rank + 1 < n && st->sa[rank + 1] > head + flen) {       /* :82 */
        const int32_t nb = st->sa[rank + 1];
        int64_t span = n - nb;
        int c;
        if (flen < span) span = flen;                                /* :85 */
        c = memcmp(st->T + head, st->T + nb, (size_t)span);           /* :86 */
        if (c < 0) break;                                            /* :88 */
        if (c == 0 && nb + flen < n && rank < st->isa[nb + flen])    /* :91-100 */
            break;
        st->sa[rank] = nb;                                           /* :104-106 */
        st->isa[nb] = rank;
        rank++;
    }
    st->sa[rank] = head;                                             /* :108-109 */
    st->isa[head] = rank;
    return rank;
}

/* mk_bwts_sa.c:133-160 -- body positions, last to first: each slides right
 * inside its first-byte bucket while its cyclic successor outranks the
 * neighbour's successor; the first position that stays put ends the pass. */
static void settle_body(fwd_state *st, int32_t head, int32_t next_head, int32_t head_rank)
{
    const int64_t n = st->n;
    int32_t follow = head_rank;
    int32_t j;
    for (j = next_head - 1; j > head; j--) {                         /* :135 */
        int32_t r = st->isa[j];
        const int32_t r0 = r;
        while (r < n - 1) {                                          /* :139 */
            const int32_t nb = st->sa[r + 1];
            if (j > nb || st->T[j] != st->T[nb] || follow < st->isa[nb + 1])  /* :141-144 */
                break;
            st->sa[r] = nb;                                          /* :148-150 */
            st->isa[nb] = r;
            r++;
        }
        st->sa[r] = j;                                               /* :152-153 */
        st->isa[j] = r;
        follow = r;                                                  /* :155 */
        if (r == r0) break;                                          /* :157-159 */
    }
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int oracle_forward_timed(const uint8_t *T, int64_t n, uint8_t *out, double phase_s[4])
{
    fwd_state st;
    int64_t i;
    int32_t low, low_at;
    double t0, t1;

    if (n <= 0 || n >= 0x7fffffffLL) return -1;
    st.T = T; st.n = n;
    st.sa = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    st.isa = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!st.sa || !st.isa) { free(st.sa); free(st.isa); return -1; }

    t0 = now_s();
    if (oracle_suffix_array(T, st.sa, n) != 0) { free(st.sa); free(st.isa); return -1; }   /* :48 */
    t1 = now_s(); if (phase_s) phase_s[0] = t1 - t0; t0 = t1;

    for (i = 0; i < n; i++) st.isa[st.sa[i]] = (int32_t)i;            /* :119-122 */
    t1 = now_s(); if (phase_s) phase_s[1] = t1 - t0; t0 = t1;

    /* :126-165 -- factor heads are the strict prefix minima of ISA */
    low = st.isa[0]; low_at = 0;
    for (i = 1; i < n && low > 0; i++) {
        if (st.isa[i] < low) {
            const int32_t hr = settle_head(&st, low_at, (int32_t)i - low_at, low);
            settle_body(&st, low_at, (int32_t)i, hr);
            low = st.isa[i];
            low_at = (int32_t)i;
        }
    }
    t1 = now_s(); if (phase_s) phase_s[2] = t1 - t0; t0 = t1;

    /* :170-188 -- bwts[isa[i]] = T[i-1]; a factor head takes its factor's last byte */
    {
        int64_t cur = n;          /* rank of the open factor's head, n = none yet */
        for (i = 0; i < n; i++) {
            if (st.isa[i] < cur) {
                if (cur < n) out[cur] = T[i - 1];
                cur = st.isa[i];
            } else {
                out[st.isa[i]] = T[i - 1];
            }
        }
        out[0] = T[n - 1];                                           /* :188 */
    }
    t1 = now_s(); if (phase_s) phase_s[3] = t1 - t0;

    free(st.sa); free(st.isa);
    return 0;
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
