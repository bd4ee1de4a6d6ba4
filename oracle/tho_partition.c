/*
 * oracle/tho_partition.c -- TEST INFRASTRUCTURE (see tho.h).
 * Stage 3: split the read into repeat units along one chain.
 * Restates get_partition_pos_with_narrow_global_alignment(), src/partition.c:171-276.
 * par_pos[i] is the index of the last base BEFORE unit i; -1 separates independent runs.
 */
#include <stdlib.h>
#include <stdio.h>
#include "tho.h"

static int imin(int a, int b) { return a < b ? a : b; }

int *tho_partition(const uint8_t *bseq, int len, const tho_chain_res_t *cr, const tho_chain_t *ch, const tho_para_t *p, int *par_n)
{
    tho_dp_t **dp = cr->dp;
#define CELL(x) (dp[ch->cell[x].i][ch->cell[x].j])
    int k = p->k, n = 0;
    int first_end = CELL(0).end, last_start = CELL(ch->len - 1).start;
    int *pos = (int *)malloc(sizeof(int) * (size_t)(len > 8 ? len : 8));
    int ch_i, s, e, s1, e1, s2, e2, n_cigar; uint32_t *cigar;

    /* left walk, src/partition.c:186-224 */
    ch_i = ch->est_ch_i; s = ch->est_start; e = ch->est_start + ch->est_period;
    while (s >= first_end && ch_i > 0) {
        s2 = s; e2 = e;
        int i;
        for (i = ch_i - 1; i >= 0; --i) {
            s1 = CELL(i).start; e1 = CELL(i).end;
            if (e1 == s) {              /* abutting anchor */
                pos[n++] = s1; ch_i = i; s = s1; e = e1;
                break;
            } else if (e1 < s) {        /* bridge [e1-k+1, e2] against [s1-k+1, s2] */
                int ql = e2 - e1 + k, tl = s2 - s1 + k;
                int iden = tho_ksw2_global_cigar(bseq + e1 - k + 1, ql, bseq + s1 - k + 1, tl, &n_cigar, &cigar, NULL);
                if (iden >= imin(tl, ql) * (1 - p->max_div)) {
                    e = s;
                    int bt = tho_ksw2_backtrack_left_end(n_cigar, cigar, ql, tl, e2 - s);
                    if (bt < 0) { fprintf(stderr, "Error: unmatched cigar and q_left_ext.\n"); exit(1); }
                    s = s2 - bt;
                    if (e == s) { ch_i = 0; free(cigar); break; }   /* no progress (the reference leaks the cigar here, :207-209) */
                    pos[n++] = s; ch_i = i + 1;
                } else {
                    pos[n++] = -1; pos[n++] = e1; pos[n++] = s1;
                    ch_i = i; s = s1; e = e1;
                }
                free(cigar);
                break;
            } else { s2 = s1; e2 = e1; }
        }
        /* The reference's for-loop can also run off i < 0 without a break (every remaining anchor
         * overlaps s); ch_i and s are then unchanged and its while-loop would spin forever.  That
         * needs s >= first_end while every earlier anchor ends after s, which contradicts
         * first_end <= s, so it cannot happen; guard anyway. */
        if (i < 0) break;
    }
    for (int a = 0, b = n - 1; a < b; ++a, --b) { int x = pos[a]; pos[a] = pos[b]; pos[b] = x; }
    pos[n++] = ch->est_start;
    pos[n++] = ch->est_start + ch->est_period;

    /* right walk, src/partition.c:236-273 */
    ch_i = ch->est_ch_i; s = ch->est_start; e = ch->est_start + ch->est_period;
    while (ch_i < ch->len - 1 && e <= last_start) {
        s1 = s; e1 = e;
        int i;
        for (i = ch_i + 1; i < ch->len; ++i) {
            s2 = CELL(i).start; e2 = CELL(i).end;
            if (s2 == e) {
                pos[n++] = e2; ch_i = i; s = s2; e = e2;
                break;
            } else if (s2 > e) {
                int ql = s2 - s1 + k, tl = e2 - e1 + k;
                int iden = tho_ksw2_global_cigar(bseq + s1 - k + 1, ql, bseq + e1 - k + 1, tl, &n_cigar, &cigar, NULL);
                if (iden >= imin(ql, tl) * (1 - p->max_div)) {
                    s = e;
                    int bt = tho_ksw2_backtrack_left_end(n_cigar, cigar, ql, tl, s2 - e);
                    if (bt < 0) { fprintf(stderr, "Error: unmatched cigar and q_left_ext.\n"); exit(1); }
                    e = e2 - bt;
                    if (e == s) { ch_i = ch->len; free(cigar); break; }
                    pos[n++] = e; ch_i = i - 1;
                } else {
                    pos[n++] = -1; pos[n++] = s2; pos[n++] = e2;
                    ch_i = i; s = s2; e = e2;
                }
                free(cigar);
                break;
            } else { s1 = s2; e1 = e2; }
        }
        if (i >= ch->len) break;
    }
#undef CELL
    *par_n = n;
    return pos;
}
