/*
 * oracle/tho_chain.c -- TEST INFRASTRUCTURE (see tho.h).
 * Stage 2: chaining DP over self-hits, score-ranked backtracking, overlap suppression.
 * Restates src/tandem_chain.c:290-404 and helpers (:14-19,:32-43,:54-111,:113-166,:170-223,:251-255).
 */
#include <stdlib.h>
#include <string.h>
#include "tho.h"

#define TOP_N 1000 /* src/tandem_chain.c:361 */

static int imin(int a, int b) { return a < b ? a : b; }
static int iabs(int a) { return a < 0 ? -a : a; }

/* floor(log2(v)); -1 for v == 0 (src/tandem_chain.c:7-19, LogTable256[0] == -1) */
static int ilog2_u32(uint32_t v)
{
    int r = -1;
    while (v) { ++r; v >>= 1; }
    return r;
}

enum { NO_CON = 0, REG_CON = 1, SAME_CON = 2, OVL_CON = 3 };

/* src/tandem_chain.c:151-166 get_con_score(); the double compare "a >= b*1.8" is kept as a
 * double compare here (the HIP path uses the integer form 5a >= 9b). */
static int con_score(const tho_dp_t *cur, const tho_dp_t *pre, int k, int *score)
{
    int cur_p = cur->end - cur->start, pre_p = pre->end - pre->start;
    if (cur->start <= pre->start || cur_p >= pre_p * 1.8 || pre_p >= cur_p * 1.8) return NO_CON;
    int dis1 = iabs(cur->end - pre->end), dis2 = iabs(cur->start - pre->start);
    int matched = imin(dis1, k) + imin(dis2, k);
    int dp = iabs(cur_p - pre_p);
    int gap = dp * dp / 2 + ilog2_u32((uint32_t)(dis1 + dis2)) / 2;
    *score = matched - gap;
    if (dp == 0) return matched < 2 * k ? OVL_CON : SAME_CON;
    return REG_CON;
}

typedef struct { int i, j, score, ord; } rank_t;

/* src/tandem_chain.c:21-23 compares score only; glibc's qsort is a stable merge sort for these
 * sizes, so ties keep enumeration order.  Made explicit with `ord`. */
static int rank_cmp(const void *a, const void *b)
{
    const rank_t *x = (const rank_t *)a, *y = (const rank_t *)b;
    if (x->score != y->score) return y->score > x->score ? 1 : -1;
    return x->ord - y->ord;
}

typedef struct { tho_cell_t *cell; int len, score; } wchain_t;

#define DP(c) (dp[(c).i][(c).j])

/* src/tandem_chain.c:54-65 */
static int overlap1(tho_dp_t **dp, const wchain_t *a, const wchain_t *b)
{
    int s1 = DP(a->cell[0]).start, e1 = DP(a->cell[a->len - 1]).start;
    int s2 = DP(b->cell[0]).start, e2 = DP(b->cell[b->len - 1]).start;
    int mn = imin(e1 - s1, e2 - s2);
    int ov = imin(e1, e2) - (s1 > s2 ? s1 : s2);
    return ov / (mn + 0.0) >= 0.5;
}

/* src/tandem_chain.c:68-83: 1 = the new chain (index ci) loses */
static int overlap_loses(tho_dp_t **dp, wchain_t *ch, int ch_n, int ci)
{
    if (ch_n <= 0 || ch[ci].len <= 0) return 0;
    int start = DP(ch[ci].cell[ch[ci].len - 1]).start;
    for (int i = ch_n - 1; i >= 0; --i) {
        if (ch[i].len <= 0) continue;
        if (DP(ch[i].cell[ch[i].len - 1]).end <= start) break;
        if (overlap1(dp, &ch[i], &ch[ci])) {
            if (ch[i].score > ch[ci].score) return 1;
            ch[i].len = 0;
            return 0;
        }
    }
    return 0;
}

/* src/tandem_chain.c:86-111 */
static int backtrack(tho_dp_t **dp, int x, int y, wchain_t *ch, int ch_n)
{
    if (dp[x][y].is_tracked) return 0;
    wchain_t *c = ch + ch_n;
    int score = dp[x][y].score, ci = x, cj = y, n = 0;
    for (;;) {
        dp[ci][cj].is_tracked = 1;
        c->cell[n].i = ci; c->cell[n].j = cj; ++n;
        int pi = dp[ci][cj].from_i, pj = dp[ci][cj].from_j;
        if (pi == -1) break;
        if (dp[pi][pj].is_tracked) { score -= dp[pi][pj].score; break; }
        ci = pi; cj = pj;
    }
    c->len = n; c->score = score;
    for (int a = 0, b = n - 1; a < b; ++a, --b) { tho_cell_t t = c->cell[a]; c->cell[a] = c->cell[b]; c->cell[b] = t; }
    return n > 1 && overlap_loses(dp, ch, ch_n, ch_n) == 0;
}

/* src/tandem_chain.c:170-185 */
static int in_chain(tho_dp_t **dp, const wchain_t *ch, const int *idx, int ch_n, int ci, int cj)
{
    int cs = dp[ci][0].start, ce = dp[ci][cj].end;
    for (int r = 0; r < ch_n; ++r) {
        const wchain_t *c = ch + idx[r];
        if (c->len <= 0) continue;
        int s = DP(c->cell[0]).start, e = DP(c->cell[c->len - 1]).end;
        if (e < cs) return 0;
        if (s > ce) continue;
        if (ce - s >= (e - s) / 2) return 1;
    }
    return 0;
}

/* src/tandem_chain.c:188-207 */
static void sort_chain(tho_dp_t **dp, const wchain_t *ch, int *idx, int ch_n)
{
    for (int a = 0; a + 1 < ch_n; ++a) {
        int i = idx[a];
        if (ch[i].len <= 0) continue;
        int e1 = DP(ch[i].cell[ch[i].len - 1]).end;
        for (int b = a + 1; b < ch_n; ++b) {
            int j = idx[b];
            if (ch[j].len <= 0) continue;
            int e2 = DP(ch[j].cell[ch[j].len - 1]).end;
            if (e1 < e2) { idx[a] = j; idx[b] = i; e1 = e2; /* i keeps its old value, as in the reference */ }
        }
    }
}

void tho_chain(const uint64_t *hits, int hit_n, const tho_para_t *p, tho_chain_res_t *res)
{
    memset(res, 0, sizeof(*res));
    if (hit_n < 2) return; /* src/tandem_chain.c:291 */
    int k = p->k, tot_n = 1;
    for (int i = 1; i < hit_n; ++i) if ((hits[i] >> 32) != (hits[i - 1] >> 32)) ++tot_n;
    int *row_size = (int *)calloc(tot_n, sizeof(int));
    tho_dp_t **dp = (tho_dp_t **)calloc(tot_n, sizeof(tho_dp_t *));
    /* rows = groups of equal end (src/tandem_chain.c:294-315), cells initialised as in init_dp :113-126 */
    for (int i = 0, r = -1; i < hit_n; ++i) {
        if (i == 0 || (hits[i] >> 32) != (hits[i - 1] >> 32)) {
            ++r;
            int n = 1;
            while (i + n < hit_n && (hits[i + n] >> 32) == (hits[i] >> 32)) ++n;
            row_size[r] = n;
            dp[r] = (tho_dp_t *)calloc(n, sizeof(tho_dp_t));
            for (int j = 0; j < n; ++j) {
                int end = (int)(hits[i + j] >> 32), period = (int)(uint32_t)hits[i + j];
                tho_dp_t d = { -1, -1, end - period, end, k + imin(k, period), 0 };
                dp[r][j] = d;
            }
        }
    }
    /* main DP, src/tandem_chain.c:325-356 */
    for (int ci = 1; ci < tot_n; ++ci) {
        for (int cj = 0; cj < row_size[ci]; ++cj) {
            tho_dp_t *cur = &dp[ci][cj];
            int best = cur->score, bi = -1, bj = -1, max_h = cur->end - cur->start, iter = 0, stop = 0;
            for (int pi = ci - 1; pi >= 0 && !stop; --pi) {
                if (dp[pi][0].end < cur->start) break;
                int gt = 0;
                for (int pj = 0; pj < row_size[pi]; ++pj) {
                    int cs, t = con_score(cur, &dp[pi][pj], k, &cs);
                    if (t == NO_CON) continue;
                    int s = dp[pi][pj].score + cs;
                    if (s > best) {
                        best = s; bi = pi; bj = pj;
                        if (t == SAME_CON || t == OVL_CON) { stop = 1; break; }
                        gt = 1;
                    } else if (t == OVL_CON) { stop = 1; break; }
                }
                if (stop) break;
                if (gt) iter = 0;
                else if (++iter >= max_h) break;
            }
            if (best > cur->score) { cur->score = best; cur->from_i = bi; cur->from_j = bj; }
        }
    }
    /* candidates: score > 0, rows descending, cells ascending (src/tandem_chain.c:32-43) */
    rank_t *rank = (rank_t *)malloc((size_t)hit_n * sizeof(rank_t));
    int rn = 0;
    for (int i = tot_n - 1; i >= 0; --i)
        for (int j = 0; j < row_size[i]; ++j)
            if (dp[i][j].score > 0) { rank_t r = { i, j, dp[i][j].score, rn }; rank[rn++] = r; }
    qsort(rank, rn, sizeof(rank_t), rank_cmp);
    /* top-N extraction, src/tandem_chain.c:361-373 */
    wchain_t *ch = (wchain_t *)calloc(TOP_N, sizeof(wchain_t));
    int *idx = (int *)malloc(TOP_N * sizeof(int));
    for (int i = 0; i < TOP_N; ++i) idx[i] = i;
    int ch_n = 0;
    for (int i = 0; i < rn && ch_n < TOP_N; ++i) {
        if (in_chain(dp, ch, idx, ch_n, rank[i].i, rank[i].j)) continue;
        if (!ch[ch_n].cell) ch[ch_n].cell = (tho_cell_t *)malloc((size_t)tot_n * sizeof(tho_cell_t));
        if (backtrack(dp, rank[i].i, rank[i].j, ch, ch_n)) ++ch_n;
        sort_chain(dp, ch, idx, ch_n);
    }
    /* output in ascending-end order, chains with >= 3 cells only (src/tandem_chain.c:393-400,:209-223,:251-255) */
    tho_chain_t *out = (tho_chain_t *)calloc(ch_n > 0 ? ch_n : 1, sizeof(tho_chain_t));
    int on = 0;
    for (int r = ch_n - 1; r >= 0; --r) {
        wchain_t *c = ch + idx[r];
        if (c->len - 1 < 2) continue; /* copy_chain: end_i - start_i < 2 (also drops len == 0) */
        out[on].cell = (tho_cell_t *)malloc((size_t)c->len * sizeof(tho_cell_t));
        memcpy(out[on].cell, c->cell, (size_t)c->len * sizeof(tho_cell_t));
        out[on].len = c->len;
        out[on].score = c->score;
        tho_dp_t d = DP(c->cell[0]);
        out[on].est_period = d.end - d.start; out[on].est_start = d.start; out[on].est_ch_i = 0;
        ++on;
    }
    for (int i = 0; i < TOP_N; ++i) free(ch[i].cell);
    free(ch); free(idx); free(rank);
    res->tot_n = tot_n; res->row_size = row_size; res->dp = dp; res->ch_n = on; res->chain = out;
}

void tho_chain_free(tho_chain_res_t *res)
{
    for (int i = 0; i < res->ch_n; ++i) free(res->chain[i].cell);
    free(res->chain);
    for (int i = 0; i < res->tot_n; ++i) free(res->dp[i]);
    free(res->dp); free(res->row_size);
    memset(res, 0, sizeof(*res));
}
