/*
 * bwts_oracle_core.h -- body of the CPU oracle's suffix sorter and forward pipeline, compiled once per index type
 * (IDX = int32_t: the reference's own width, mk_bwts_sa.c:26-27; IDX = int64_t: inputs of 2^31 bytes and more, which the
 * reference cannot index).  TEST INFRASTRUCTURE ONLY (see bwts_oracle.h).  Included by bwts_oracle.c with IDX and N() set.
 */
typedef struct {
    const uint8_t *b;   /* level-0 bytes, or NULL */
    const IDX *w;   /* deeper-level symbols, or NULL */
    int64_t nb;         /* number of real bytes at level 0 */
} N(sym_src);

static inline IDX N(sym_at)(const N(sym_src) *s, int64_t i)
{
    if (s->w) return s->w[i];
    return i < s->nb ? (IDX)s->b[i] + 1 : 0;
}

#define TY_GET(i)  ((ty[(i) >> 3] >> ((i) & 7)) & 1)          /* 1 = S-type */
#define TY_SET(i)  (ty[(i) >> 3] |= (uint8_t)(1u << ((i) & 7)))
#define IS_LMS(i)  ((i) > 0 && TY_GET(i) && !TY_GET((i) - 1))

static void N(bucket_bounds)(const N(sym_src) *s, int64_t n, IDX K, IDX *bkt, int ends)
{
    int64_t i;
    IDX sum = 0;
    for (i = 0; i < K; i++) bkt[i] = 0;
    for (i = 0; i < n; i++) bkt[N(sym_at)(s, i)]++;
    for (i = 0; i < K; i++) {
        sum += bkt[i];
        bkt[i] = ends ? sum : sum - bkt[i];
    }
}

static void N(induce_l)(const N(sym_src) *s, const uint8_t *ty, IDX *SA, int64_t n, IDX K, IDX *bkt)
{
    int64_t i;
    N(bucket_bounds)(s, n, K, bkt, 0);
    for (i = 0; i < n; i++) {
        int64_t j = (int64_t)SA[i] - 1;
        if (j >= 0 && !TY_GET(j)) SA[bkt[N(sym_at)(s, j)]++] = (IDX)j;
    }
}

static void N(induce_s)(const N(sym_src) *s, const uint8_t *ty, IDX *SA, int64_t n, IDX K, IDX *bkt)
{
    int64_t i;
    N(bucket_bounds)(s, n, K, bkt, 1);
    for (i = n - 1; i >= 0; i--) {
        int64_t j = (int64_t)SA[i] - 1;
        if (j >= 0 && TY_GET(j)) SA[--bkt[N(sym_at)(s, j)]] = (IDX)j;
    }
}

static int N(sais_level)(const N(sym_src) *s, IDX *SA, int64_t n, IDX K)
{
    int64_t i, j, n1;
    IDX name, prev;
    uint8_t *ty = (uint8_t *)calloc((size_t)(n / 8 + 1), 1);
    IDX *bkt = (IDX *)malloc(sizeof(IDX) * (size_t)K);
    if (!ty || !bkt) { free(ty); free(bkt); return -1; }

    /* classify: last symbol (terminator) is S, the one before it is L */
    TY_SET(n - 1);
    for (i = n - 3; i >= 0; i--) {
        IDX a = N(sym_at)(s, i), b = N(sym_at)(s, i + 1);
        if (a < b || (a == b && TY_GET(i + 1))) TY_SET(i);
    }

    /* stage 1: sort LMS substrings */
    N(bucket_bounds)(s, n, K, bkt, 1);
    for (i = 0; i < n; i++) SA[i] = -1;
    for (i = 1; i < n; i++)
        if (IS_LMS(i)) SA[--bkt[N(sym_at)(s, i)]] = (IDX)i;
    N(induce_l)(s, ty, SA, n, K, bkt);
    N(induce_s)(s, ty, SA, n, K, bkt);

    n1 = 0;
    for (i = 0; i < n; i++)
        if (IS_LMS(SA[i])) SA[n1++] = SA[i];
    for (i = n1; i < n; i++) SA[i] = -1;

    name = 0; prev = -1;
    for (i = 0; i < n1; i++) {
        IDX pos = SA[i];
        int diff = 0;
        int64_t d;
        for (d = 0; d < n; d++) {
            if (prev == -1 || N(sym_at)(s, pos + d) != N(sym_at)(s, prev + d) ||
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
        IDX *SA1 = SA, *s1 = SA + n - n1;
        if (name < n1) {
            N(sym_src) sub; sub.b = NULL; sub.w = s1; sub.nb = 0;
            if (N(sais_level)(&sub, SA1, n1, name) != 0) { free(ty); free(bkt); return -1; }
        } else {
            for (i = 0; i < n1; i++) SA1[s1[i]] = (IDX)i;
        }

        /* stage 3: induce the full order from the sorted LMS suffixes */
        N(bucket_bounds)(s, n, K, bkt, 1);
        for (i = 1, j = 0; i < n; i++)
            if (IS_LMS(i)) s1[j++] = (IDX)i;
        for (i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
        for (i = n1; i < n; i++) SA[i] = -1;
        for (i = n1 - 1; i >= 0; i--) {
            j = SA[i]; SA[i] = -1;
            SA[--bkt[N(sym_at)(s, j)]] = (IDX)j;
        }
    }
    N(induce_l)(s, ty, SA, n, K, bkt);
    N(induce_s)(s, ty, SA, n, K, bkt);

    free(ty); free(bkt);
    return 0;
}

/* buf has n + 1 entries; on return buf[0] is the terminator suffix and buf + 1 the suffix array (shorter-first ties) */
static int N(suffix_array_in)(const uint8_t *T, IDX *buf, int64_t n)
{
    N(sym_src) s;
    if (n == 1) { buf[0] = 1; buf[1] = 0; return 0; }
    s.b = T; s.w = NULL; s.nb = n;
    return N(sais_level)(&s, buf, n + 1, 257);
}

static int N(suffix_array)(const uint8_t *T, IDX *SA, int64_t n)
{
    IDX *tmp;
    if (n <= 0) return 0;
    tmp = (IDX *)malloc(sizeof(IDX) * (size_t)(n + 1));
    if (!tmp) return -1;
    if (N(suffix_array_in)(T, tmp, n) != 0) { free(tmp); return -1; }
    memcpy(SA, tmp + 1, sizeof(IDX) * (size_t)n);
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Forward transform, reference pipeline                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t *T;
    int64_t n;
    IDX *sa, *isa;
} N(fwd_state);

/* mk_bwts_sa.c:74-112 -- a factor's head moves from its suffix rank to its
 * rotation rank: it slides right past every later-in-text suffix that the
 * factor (repeated) is not smaller than. */
static IDX N(settle_head)(N(fwd_state) *st, IDX head, IDX flen, IDX rank)
{
    const int64_t n = st->n;
    while (rank + 1 < n && st->sa[rank + 1] > head + flen) {       /* :82 */
        const IDX nb = st->sa[rank + 1];
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
static void N(settle_body)(N(fwd_state) *st, IDX head, IDX next_head, IDX head_rank)
{
    const int64_t n = st->n;
    IDX follow = head_rank;
    IDX j;
    for (j = next_head - 1; j > head; j--) {                         /* :135 */
        IDX r = st->isa[j];
        const IDX r0 = r;
        while (r < n - 1) {                                          /* :139 */
            const IDX nb = st->sa[r + 1];
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

static int N(forward_timed)(const uint8_t *T, int64_t n, uint8_t *out, double phase_s[4])
{
    N(fwd_state) st;
    int64_t i;
    IDX low, low_at;
    IDX *sa_buf;
    double t0, t1;

    if (n <= 0 || (sizeof(IDX) == 4 && n >= 0x7fffffffLL)) return -1;
    st.T = T; st.n = n;
    sa_buf = (IDX *)malloc(sizeof(IDX) * (size_t)(n + 1));
    st.isa = (IDX *)malloc(sizeof(IDX) * (size_t)n);
    if (!sa_buf || !st.isa) { free(sa_buf); free(st.isa); return -1; }
    st.sa = sa_buf + 1;

    t0 = now_s();
    if (N(suffix_array_in)(T, sa_buf, n) != 0) { free(sa_buf); free(st.isa); return -1; }   /* :48 */
    t1 = now_s(); if (phase_s) phase_s[0] = t1 - t0; t0 = t1;

    for (i = 0; i < n; i++) st.isa[st.sa[i]] = (IDX)i;            /* :119-122 */
    t1 = now_s(); if (phase_s) phase_s[1] = t1 - t0; t0 = t1;

    /* :126-165 -- factor heads are the strict prefix minima of ISA */
    low = st.isa[0]; low_at = 0;
    for (i = 1; i < n && low > 0; i++) {
        if (st.isa[i] < low) {
            const IDX hr = N(settle_head)(&st, low_at, (IDX)i - low_at, low);
            N(settle_body)(&st, low_at, (IDX)i, hr);
            low = st.isa[i];
            low_at = (IDX)i;
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

    free(sa_buf); free(st.isa);
    return 0;
}

