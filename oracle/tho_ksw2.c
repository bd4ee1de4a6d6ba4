/*
 * oracle/tho_ksw2.c -- TEST INFRASTRUCTURE (see tho.h).
 * Restatement of the three ksw2 modes TideHunter uses (src/ksw2_align.c:117-173):
 *   global + CIGAR (flag 0), and extension score-only (EXTZ_ONLY|SCORE_ONLY), always with
 *   w = -1 and zdrop = -1, i.e. the full qlen x tlen matrix (ksw2/ksw2_extz2_sse.c:72).
 * ksw2 computes a Suzuki-Kasahara difference recurrence in 8-bit lanes; it is exact, so this
 * file states the same thing as a plain Gotoh recurrence in 32-bit integers:
 *   H(-1,-1)=0, H(i,-1) = -(q+(i+1)e), H(-1,j) = -(q+(j+1)e)      (boundary: :111-116)
 *   E(i,j) = max(H(i-1,j)-q, E(i-1,j)) - e   (gap that consumes target = CIGAR 'D', op 2)
 *   F(i,j) = max(H(i,j-1)-q, F(i,j-1)) - e   (gap that consumes query  = CIGAR 'I', op 1)
 *   H(i,j) = max(H(i-1,j-1)+s, E, F)
 * with i over target, j over query, s = +1 / -2, or -e = -1 when either code is 4
 * (src/ksw2_align.c:12-17; ksw2_extz2_sse.c:68,130; codes > 4 compare by equality).
 * Pinned against the compiled reference in tests/test_oracle_ref.py (oracle/_ref).
 */
#include <stdlib.h>
#include <string.h>
#include "tho.h"

#define GAPO 2
#define GAPE 1
#define NEG_INF (-0x40000000)

static inline int sc(uint8_t a, uint8_t b)
{
    if (a == 4 || b == 4) return -GAPE;
    return a == b ? 1 : -2;
}

static uint32_t *push_cigar(uint32_t *cg, int *n, int *m, uint32_t op, int len)
{
    if (*n == 0 || op != (cg[*n - 1] & 0xf)) {
        if (*n == *m) { *m = *m ? *m << 1 : 4; cg = (uint32_t *)realloc(cg, (size_t)*m * 4); }
        cg[(*n)++] = (uint32_t)len << 4 | op;
    } else cg[*n - 1] += (uint32_t)len << 4;
    return cg;
}

/* Traceback byte per cell, as written at ksw2_extz2_sse.c:171-196:
 *   low bits: 0 if the diagonal gives H, 1 if E does (strictly better than diagonal),
 *             2 if F does (strictly better than both)  => ties prefer diagonal, then E, then F;
 *   0x08: extending E into the next row is strictly better than re-opening from H;
 *   0x10: likewise for F. */
int tho_ksw2_global_cigar(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int *n_cigar, uint32_t **cigar, int *score)
{
    *n_cigar = 0; *cigar = NULL; if (score) *score = NEG_INF;
    if (qlen <= 0 || tlen <= 0) return 0; /* ksw2_extz2_sse.c:57; ksw2_get_xid of an empty cigar = 0 */
    /* the traceback matrix (1 MB at 1000 x 1000, 25 MB at 5000 x 5000) is per-thread and grow-only: see tho_poa.c */
    static __thread uint8_t *tl_p; static __thread size_t tl_cap;
    if ((size_t)qlen * tlen > tl_cap) { tl_cap = (size_t)qlen * tlen + ((size_t)qlen * tlen >> 2); free(tl_p); tl_p = (uint8_t *)malloc(tl_cap); }
    uint8_t *p = tl_p;
    int *H = (int *)malloc(sizeof(int) * qlen), *E = (int *)malloc(sizeof(int) * qlen);
    for (int j = 0; j < qlen; ++j) {
        H[j] = -(GAPO + (j + 1) * GAPE);            /* H(-1,j) */
        E[j] = H[j] - GAPO - GAPE;                  /* E(0,j): open from the boundary */
    }
    for (int i = 0; i < tlen; ++i) {
        int hdiag = i == 0 ? 0 : -(GAPO + i * GAPE);      /* H(i-1,-1) */
        int hleft = -(GAPO + (i + 1) * GAPE);             /* H(i,-1) */
        int f = hleft - GAPO - GAPE;                      /* F(i,0) */
        uint8_t *pr = p + (size_t)i * qlen;
        for (int j = 0; j < qlen; ++j) {
            int diag = hdiag + sc(t[i], q[j]);
            int e = E[j];
            uint8_t d = e > diag ? 1 : 0;
            int h = e > diag ? e : diag;
            if (f > h) { d = 2; h = f; }
            if (e > h - GAPO) d |= 0x08;
            if (f > h - GAPO) d |= 0x10;
            pr[j] = d;
            hdiag = H[j];
            H[j] = h;
            E[j] = (e > h - GAPO ? e : h - GAPO) - GAPE;   /* E(i+1,j) */
            f = (f > h - GAPO ? f : h - GAPO) - GAPE;      /* F(i,j+1) */
        }
    }
    if (score) *score = H[qlen - 1];
    /* ksw_backtrack, ksw2/ksw2.h:119-151, from (tlen-1, qlen-1); the band is full so no state is forced */
    int n = 0, m = 0, i = tlen - 1, j = qlen - 1, state = 0;
    uint32_t *cg = NULL;
    while (i >= 0 && j >= 0) {
        uint8_t tmp = p[(size_t)i * qlen + j];
        if (state == 0) state = tmp & 7;
        else if (!((tmp >> (state + 2)) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (state == 0) { cg = push_cigar(cg, &n, &m, 0, 1); --i; --j; }
        else if (state == 1) { cg = push_cigar(cg, &n, &m, 2, 1); --i; }
        else { cg = push_cigar(cg, &n, &m, 1, 1); --j; }
    }
    if (i >= 0) cg = push_cigar(cg, &n, &m, 2, i + 1);
    if (j >= 0) cg = push_cigar(cg, &n, &m, 1, j + 1);
    for (int a = 0; a < n >> 1; ++a) { uint32_t x = cg[a]; cg[a] = cg[n - 1 - a]; cg[n - 1 - a] = x; }
    free(H); free(E);
    /* src/ksw2_align.c:62-86: count M columns with equal codes (N vs N counts) */
    int iden = 0, qi = 0, ti = 0;
    for (int a = 0; a < n; ++a) {
        int op = cg[a] & 0xf, len = (int)(cg[a] >> 4);
        if (op == 0) { for (int x = 0; x < len; ++x) iden += q[qi + x] == t[ti + x]; qi += len; ti += len; }
        else if (op == 1) qi += len;
        else ti += len;
    }
    *n_cigar = n; *cigar = cg;
    return iden;
}

int tho_ksw2_global(const uint8_t *q, int qlen, const uint8_t *t, int tlen)
{
    int n; uint32_t *cg;
    int iden = tho_ksw2_global_cigar(q, qlen, t, tlen, &n, &cg, NULL);
    free(cg);
    return iden;
}

/* src/ksw2_align.c:88-115: target bases consumed by the alignment's last q_left_ext query bases.
 * Returns -1 where the reference exits with "unmatched cigar and q_left_ext". */
int tho_ksw2_backtrack_left_end(int n_cigar, const uint32_t *cigar, int qlen, int tlen, int q_left_ext)
{
    (void)qlen; (void)tlen;
    int t_ext = 0, q_rem = q_left_ext;
    for (int i = n_cigar - 1; i >= 0; --i) {
        int op = cigar[i] & 0xf, len = (int)(cigar[i] >> 4);
        if (op == 0) {
            if (len >= q_rem) return t_ext + q_rem;
            t_ext += len; q_rem -= len;
        } else if (op == 1) {
            if (len >= q_rem) return t_ext;
            q_rem -= len;
        } else if (op == 2) t_ext += len;
    }
    if (q_rem > 0) return -1;
    return t_ext;
}

/* Extension, score only (ksw2_extz2_sse.c:224-269 + ksw_apply_zdrop ksw2.h:160-176 with zdrop<0):
 * anti-diagonals r = 0,1,...; the first strictly greater H over all of them wins (max starts
 * at 0).  Within one anti-diagonal the candidate order is an artefact of the SSE code:
 * first t = en0, then four interleaved lanes over [st0, en1) each keeping its first strict
 * maximum and merged in lane order with strict '>', then the scalar tail [en1, en0). */
void tho_ksw2_right_ext(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int *max_q, int *max_t)
{
    *max_q = *max_t = -1;
    if (qlen <= 0 || tlen <= 0) return;
    /* anti-diagonal sweep with O(tlen) memory: X_r[i] = X(i, r-i); buffers rotate by pointer */
    int *buf = (int *)malloc(sizeof(int) * (size_t)tlen * 7);
    int *Hc = buf, *H1 = buf + tlen, *H2 = buf + 2 * tlen;
    int *Ec = buf + 3 * tlen, *E1 = buf + 4 * tlen, *Fc = buf + 5 * tlen, *F1 = buf + 6 * tlen;
    int best = 0;
    for (int r = 0; r < qlen + tlen - 1; ++r) {
        int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = r < tlen - 1 ? r : tlen - 1;
        for (int i = st0; i <= en0; ++i) {
            int j = r - i;
            /* neighbours on previous anti-diagonals (or boundaries) */
            int hdiag = (i == 0 && j == 0) ? 0 : i == 0 ? -(GAPO + j * GAPE) : j == 0 ? -(GAPO + i * GAPE) : H2[i - 1];
            int hup = i == 0 ? -(GAPO + (j + 1) * GAPE) : H1[i - 1];   /* H(i-1,j) */
            int hleft = j == 0 ? -(GAPO + (i + 1) * GAPE) : H1[i];     /* H(i,j-1) */
            int e = i == 0 ? hup - GAPO - GAPE : (hup - GAPO > E1[i - 1] ? hup - GAPO : E1[i - 1]) - GAPE;
            int f = j == 0 ? hleft - GAPO - GAPE : (hleft - GAPO > F1[i] ? hleft - GAPO : F1[i]) - GAPE;
            int h = hdiag + sc(t[i], q[j]);
            if (e > h) h = e;
            if (f > h) h = f;
            Hc[i] = h; Ec[i] = e; Fc[i] = f;
        }
        int mH, mT;
        if (r > 0) {
            mH = Hc[en0]; mT = en0;
            int en1 = st0 + (en0 - st0) / 4 * 4, lh[4], lt[4];
            for (int l = 0; l < 4; ++l) { lh[l] = mH; lt[l] = en0; }
            for (int x = st0; x < en1; x += 4)
                for (int l = 0; l < 4; ++l)
                    if (Hc[x + l] > lh[l]) { lh[l] = Hc[x + l]; lt[l] = x; }
            for (int l = 0; l < 4; ++l) if (mH < lh[l]) { mH = lh[l]; mT = lt[l] + l; }
            for (int x = en1; x < en0; ++x) if (Hc[x] > mH) { mH = Hc[x]; mT = x; }
        } else { mH = Hc[0]; mT = 0; }
        if (mH > best) { best = mH; *max_t = mT; *max_q = r - mT; }
        int *x = H2; H2 = H1; H1 = Hc; Hc = x;
        x = E1; E1 = Ec; Ec = x;
        x = F1; F1 = Fc; Fc = x;
    }
    free(buf);
}

/* src/ksw2_align.c:161-173: reverse both, then extend */
void tho_ksw2_left_ext(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int *max_q, int *max_t)
{
    uint8_t *rq = (uint8_t *)malloc(qlen > 0 ? qlen : 1), *rt = (uint8_t *)malloc(tlen > 0 ? tlen : 1);
    for (int i = 0; i < qlen; ++i) rq[i] = q[qlen - 1 - i];
    for (int i = 0; i < tlen; ++i) rt[i] = t[tlen - 1 - i];
    tho_ksw2_right_ext(rq, qlen, rt, tlen, max_q, max_t);
    free(rq); free(rt);
}
