/*
 * oracle/tho_edlib.c -- TEST INFRASTRUCTURE (see tho.h).
 * Restates edlib_align_HW(), src/edlib_align.c:73-85: edlibAlign() in EDLIB_MODE_HW with
 * EDLIB_TASK_LOC (edlib/src/edlib.cpp:141-275) -- the infix edit distance of the query in the
 * target if it is <= k (else -1), the FIRST target position where that distance is reached
 * (endLocations[0]; positions are collected in ascending order, :640-655,:666-677) and the
 * start edlib derives for it by running the reversed query against the reversed target
 * prefix in SHW mode and taking the LAST best position (:214-226), i.e. the smallest start s
 * such that the global distance between the query and target[s..end] equals the optimum.
 * edlib's Myers bit-vector + Ukkonen band is exact for scores <= k, so a plain O(mn) DP states
 * the same thing.  Equality is byte equality plus a/A c/C g/G t/T n/N (src/edlib_align.c:21-27).
 * k < 0 means "no limit" (edlib.cpp:171-191).
 */
#include <stdlib.h>
#include "tho.h"

static int eq(char a, char b)
{
    if (a == b) return 1;
    char x = a | 0x20, y = b | 0x20;   /* fold case */
    if (x != y) return 0;
    /* only a,c,g,t,n are case-equivalent; the two bytes must be exactly the lower/upper pair */
    if ((a ^ b) != 0x20) return 0;
    return x == 'a' || x == 'c' || x == 'g' || x == 't' || x == 'n';
}

int tho_edlib_hw(const char *q, int qlen, const char *t, int tlen, int *start, int *end, int k)
{
    if (qlen <= 0 || tlen <= 0) return -1; /* not reached by TideHunter's callers with real adapters */
    int *col = (int *)malloc(sizeof(int) * (size_t)(qlen + 1));
    /* HW: D(0,c) = 0 for every column (free leading target), D(i,-1) = i */
    for (int i = 0; i <= qlen; ++i) col[i] = i;
    int best = -1, best_end = -1;
    for (int c = 0; c < tlen; ++c) {
        int diag = col[0];
        col[0] = 0;
        for (int i = 1; i <= qlen; ++i) {
            int up = col[i - 1] + 1, left = col[i] + 1, d = diag + !eq(q[i - 1], t[c]);
            diag = col[i];
            int v = d < up ? d : up;
            col[i] = v < left ? v : left;
        }
        if (best < 0 || col[qlen] < best) { best = col[qlen]; best_end = c; }
    }
    if (k >= 0 && best > (k < qlen ? k : qlen)) { free(col); return -1; }
    /* edlib pads the query to a multiple of 64 rows with W wildcard rows and reads column c-W's
     * score in column c (:640-655); when W > 0 the score of "position -1" (query entirely before
     * the target, score qlen) is therefore also seen, first.  It can only be optimal when nothing
     * better exists (best == qlen, which needs k >= qlen); edlib then reports end -1 and start 0
     * (edlib.cpp:200-213). */
    if (best >= qlen && (64 - qlen % 64) % 64 > 0) { *start = 0; *end = -1; free(col); return qlen; }
    /* start: reversed query vs reversed target[0..best_end], SHW (leading target gaps cost), last best position */
    int n = best_end + 1;
    for (int i = 0; i <= qlen; ++i) col[i] = i;
    int last_p = -1;
    for (int c = 0; c < n; ++c) {
        int diag = col[0];
        col[0] = c + 1;
        for (int i = 1; i <= qlen; ++i) {
            int up = col[i - 1] + 1, left = col[i] + 1, d = diag + !eq(q[qlen - i], t[best_end - c]);
            diag = col[i];
            int v = d < up ? d : up;
            col[i] = v < left ? v : left;
        }
        if (col[qlen] == best) last_p = c;
    }
    free(col);
    *end = best_end;
    *start = best_end - last_p;
    return best;
}
