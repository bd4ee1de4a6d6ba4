/*
 * oracle/tho_cons.c -- TEST INFRASTRUCTURE (see tho.h).
 * Stage 4 host logic and the per-read driver.
 * Restates src/tidehunter.c:23-60, src/gen_cons.c:10-62,89-301, src/abpoa_cons.c:30-120.
 *
 * Deliberate deviations from latent reference bugs:
 *  - src/gen_cons.c:206 spins forever when POA yields an empty consensus; here the run is skipped.
 *  - cons_seq->qual.l is never reset in the reference (src/main.c:266-268), which corrupts
 *    FASTQ / -f 4 qualities from the second 4096-read chunk on and after a -l replacement;
 *    here every record owns its quality string.
 *  Neither affects default FASTA / tabular output.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <limits.h>
#include "tho.h"

void tho_read_res_clear(tho_read_res_t *r)
{
    for (int i = 0; i < r->n; ++i) { free(r->a[i].seq); free(r->a[i].qual); free(r->a[i].sub_pos); }
    r->n = 0;
}
void tho_read_res_free(tho_read_res_t *r) { tho_read_res_clear(r); free(r->a); r->a = NULL; r->m = 0; }

static tho_cons_t *res_push(tho_read_res_t *r)
{
    if (r->n == r->m) { r->m = r->m ? r->m * 2 : 2; r->a = (tho_cons_t *)realloc(r->a, sizeof(tho_cons_t) * r->m); }
    tho_cons_t *c = &r->a[r->n++];
    memset(c, 0, sizeof(*c));
    return c;
}

/* src/gen_cons.c:10-62 write_tandem_cons_seq() */
static tho_cons_t *write_cons(tho_read_res_t *r, const char *seq, const uint8_t *qual, int cons_len, int start, int end,
                              double copy_num, double ave_match, const tho_para_t *p, int full_length, const int *pos, int pos_n)
{
    if (cons_len < p->min_len || cons_len > p->max_p) return NULL;
    if (p->only_longest && r->n == 1) {
        if (end - start > r->a[0].end - r->a[0].start) tho_read_res_clear(r);
        else return NULL;
    }
    tho_cons_t *c = res_push(r);
    c->seq = (char *)malloc((size_t)cons_len + 1); memcpy(c->seq, seq, cons_len); c->seq[cons_len] = 0;
    if (qual) { c->qual = (char *)malloc((size_t)cons_len + 1); memcpy(c->qual, qual, cons_len); c->qual[cons_len] = 0; }
    c->cons_len = cons_len; c->start = start; c->end = end; c->copy_num = copy_num; c->ave_match = ave_match;
    c->full_length = full_length;
    c->sub_pos = (int *)malloc(sizeof(int) * (size_t)pos_n); memcpy(c->sub_pos, pos, sizeof(int) * (size_t)pos_n); c->pos_n = pos_n;
    return c;
}

/* src/gen_cons.c:64-83 write_tandem_unit() */
static void write_unit(tho_read_res_t *r, const int *pos, int pos_n)
{
    tho_cons_t *c = res_push(r);
    c->sub_pos = (int *)malloc(sizeof(int) * (size_t)pos_n); memcpy(c->sub_pos, pos, sizeof(int) * (size_t)pos_n); c->pos_n = pos_n;
}

/* src/abpoa_cons.c:30-120 abpoa_gen_cons(): returns cons_len (0 = nothing to report) */
static int gen_cons(const uint8_t *bseq, int seq_len, const int *pos, int pos_n, uint8_t *cons, uint8_t *qual, const tho_para_t *p, int *n_seqs_out)
{
    int n = 0, cons_len = 0;
    int *lens = (int *)malloc(sizeof(int) * (size_t)pos_n);
    const uint8_t **seqs = (const uint8_t **)malloc(sizeof(uint8_t *) * (size_t)pos_n);
    for (int i = 0; i < pos_n - 1; ++i) {
        int start = pos[i], end = pos[i + 1];
        if (start < 0 || end < 0 || start >= seq_len - 1 || end + 1 > seq_len) continue;   /* :43 */
        lens[n] = end - start; seqs[n] = bseq + start + 1; ++n;
    }
    *n_seqs_out = n;
    int min_cov = 0;
    if (p->min_frac > 0.0) min_cov = (int)(n * p->min_frac);
    else if (p->min_cov > 0) min_cov = p->min_cov;
    if (n <= 2) {
        if (n <= 1) { fprintf(stderr, "Not enough sequences to perform msa.\n"); exit(EXIT_FAILURE); } /* :61 */
        cons_len = lens[0];
        int skip = 0;
        if (min_cov > 0) {
            int mc = 2;
            if (lens[0] != lens[1]) mc = 1;
            else for (int i = 0; i < cons_len; ++i) if (seqs[0][i] != seqs[1][i]) { mc = 1; break; }
            if (mc < min_cov) skip = 1;
        }
        if (!skip) for (int i = 0; i < cons_len; ++i) { cons[i] = seqs[0][i]; if (qual) qual[i] = 33; }
        else cons_len = 0;
    } else {
        int tot = 0; for (int i = 0; i < n; ++i) tot += lens[i];
        int *cov = (int *)malloc(sizeof(int) * (size_t)(tot + 1));
        uint8_t *tmp = (uint8_t *)malloc((size_t)tot + 1);
        cons_len = tho_poa_consensus(n, seqs, lens, p, tmp, cov);
        int skip = 0;
        if (min_cov > 0) for (int i = 0; i < cons_len; ++i) if (cov[i] < min_cov) { skip = 1; break; }
        if (qual) for (int i = 0; i < cons_len; ++i) {            /* :100-107 */
            double x = 13.8 * (1.25 * cov[i] / n - 0.25);
            double pe = 1 - 1.0 / (1.0 + pow(2.718281828459045, -1 * x));
            qual[i] = (uint8_t)(33 + (int)(-10 * log10(pe) + 0.499));
        }
        if (!skip) memcpy(cons, tmp, (size_t)cons_len); else cons_len = 0;
        free(cov); free(tmp);
    }
    free(lens); free(seqs);
    return cons_len;
}

/* src/gen_cons.c:173-301 seqs_msa() */
static void seqs_msa(int seq_len, const uint8_t *bseq, int par_n, const int *par_pos, tho_read_res_t *out, const tho_para_t *p)
{
    char *cons_seq = (char *)malloc((size_t)seq_len + 1);
    uint8_t *cons_b = (uint8_t *)malloc((size_t)seq_len + 1), *cons_q = NULL;
    if (p->out_fmt == THO_FASTQ_FMT || p->out_fmt == THO_TAB_QUAL_FMT) cons_q = (uint8_t *)malloc((size_t)seq_len + 1);
    int i = 0, j;
    while (i < par_n - p->min_copy) {
        if (par_pos[i] < 0) { ++i; continue; }
        for (j = i + 1; j < par_n; ++j) if (par_pos[j] < 0) break;
        if (j - i > p->min_copy) {
            if (p->only_unit) write_unit(out, par_pos + i, j - i);
            else {
                int n_seqs, cons_len = gen_cons(bseq, seq_len, par_pos + i, j - i, cons_b, cons_q, p, &n_seqs);
                if (cons_len == 0) { i = j + 1; continue; }        /* deviation: reference loops forever, :206 */
                double ave = 0;
                for (int k = i; k < j - 1; ++k) {                    /* :209-214 */
                    int start = par_pos[k], len = par_pos[k + 1] - start;
                    int iden = tho_ksw2_global(bseq + start + 1, len, cons_b, cons_len);
                    ave += (iden * 100 / (len + 0.0));
                }
                for (int s = 0; s < cons_len; ++s) cons_seq[s] = cons_b[s] <= 4 ? "ACGTN"[cons_b[s]] : 'N';
                cons_seq[cons_len] = 0;
                int mq, mt, mql, mtl;
                double copy_num = n_seqs;
                tho_ksw2_left_ext(cons_b, cons_len, bseq, par_pos[i] + 1, &mql, &mtl);           /* :218 */
                int cons_start = par_pos[i] - mtl;
                copy_num += (mql + 1.0) / cons_len;
                tho_ksw2_right_ext(cons_b, cons_len, bseq + par_pos[j - 1] + 1, seq_len - par_pos[j - 1] - 1, &mq, &mt); /* :221 */
                int cons_end = par_pos[j - 1] + mt + 1;
                copy_num += (mq + 1.0) / cons_len;
                int full_length = 0, out_len = cons_len;
                if (p->five_seq && p->three_seq && cons_len > p->five_len + p->three_len) {       /* :226-291 */
                    char *cons2 = (char *)malloc(((size_t)cons_len << 1) + 1);
                    uint8_t *qual2 = NULL;
                    memcpy(cons2, cons_seq, cons_len); memcpy(cons2 + cons_len, cons_seq, cons_len); cons2[cons_len << 1] = 0;
                    if (cons_q) { qual2 = (uint8_t *)malloc((size_t)cons_len << 1); memcpy(qual2, cons_q, cons_len); memcpy(qual2 + cons_len, cons_q, cons_len); }
                    int tar_start = -1, tar_end = -1, tot_ed = INT_MAX, e5, e3, s5 = -1, n5 = -1, s3 = -1, n3 = -1;
                    int k5 = (int)(p->five_len * (1 - p->ada_match_rat)), k3 = (int)(p->three_len * (1 - p->ada_match_rat));
                    e5 = tho_edlib_hw(p->five_seq, p->five_len, cons2, cons_len << 1, &s5, &n5, k5);
                    if (e5 == -1) goto REV;
                    e3 = tho_edlib_hw(p->three_rc_seq, p->three_len, cons2, cons_len << 1, &s3, &n3, k3);
                    if (e3 == -1) goto REV;
                    if (s3 <= n5) {
                        if (n3 + cons_len < cons_len << 1 && s3 + cons_len > n5) {
                            tar_start = n5 + 1; tar_end = s3 + cons_len - 1; full_length = 1; tot_ed = e5 + e3;
                        }
                    } else { tar_start = n5 + 1; tar_end = s3 - 1; tot_ed = e5 + e3; full_length = 1; }
                    if (tot_ed == 0) goto WRITE;
REV:
                    e5 = tho_edlib_hw(p->five_rc_seq, p->five_len, cons2, cons_len << 1, &s5, &n5, k5);
                    if (e5 == -1) goto WRITE;
                    e3 = tho_edlib_hw(p->three_seq, p->three_len, cons2, cons_len << 1, &s3, &n3, k3);
                    if (e3 == -1) goto WRITE;
                    if (e5 + e3 < tot_ed) {
                        if (s5 <= n3) {
                            if (n5 + cons_len < cons_len << 1 && s5 + cons_len > n3) { tar_start = n3 + 1; tar_end = s5 + cons_len - 1; full_length = 2; }
                        } else { tar_start = n3 + 1; tar_end = s5 - 1; full_length = 2; }
                    }
WRITE:
                    if (tar_start > 0 && tar_end > tar_start) {
                        out_len = tar_end - tar_start + 1;
                        memcpy(cons_seq, cons2 + tar_start, out_len); cons_seq[out_len] = 0;
                        if (cons_q) for (int k = tar_start; k <= tar_end; ++k) cons_q[k - tar_start] = qual2[k];
                    }
                    free(cons2); free(qual2);
                }
                if (!p->only_full_length || full_length > 0) {
                    tho_cons_t *c = write_cons(out, cons_seq, cons_q, out_len, cons_start, cons_end, copy_num, ave / (j - i - 1), p, full_length, par_pos + i, j - i);
                    if (c) { c->n_seqs = n_seqs; c->max_q_l = mql; c->max_t_l = mtl; c->max_q_r = mq; c->max_t_r = mt; }
                }
            }
        }
        i = j + 1;
    }
    free(cons_seq); free(cons_b); free(cons_q);
}

typedef struct { int ed, start, end; } ed_res_t;

/* src/gen_cons.c:89-110 */
static int collect_ed(const tho_para_t *p, const char *q, int qlen, const char *seq, int seq_len, ed_res_t *r)
{
    int n = 0, ed, s, e, k = (int)(qlen * (1 - p->ada_match_rat));
    ed = tho_edlib_hw(q, qlen, seq, seq_len, &s, &e, k);
    if (ed != -1) {
        r[0].ed = ed; r[0].start = s; r[0].end = e; n = 1;
        if (r[0].start >= qlen) {
            ed = tho_edlib_hw(q, qlen, seq, r[0].start, &s, &e, k);
            if (ed != -1) { r[n].ed = ed; r[n].start = s; r[n].end = e; ++n; }
        }
        if (r[0].end <= seq_len - qlen) {
            ed = tho_edlib_hw(q, qlen, seq + r[0].end, seq_len - r[0].end, &s, &e, k);
            if (ed != -1) { r[n].ed = ed; r[n].start = r[0].end + s; r[n].end = r[0].end + e; ++n; }
        }
    }
    return n;
}

/* src/gen_cons.c:112-126 */
static int full_len_pick(const tho_para_t *p, int ln, const ed_res_t *l, int rn, const ed_res_t *r, int *ts, int *te)
{
    int tot = INT_MAX;
    for (int i = 0; i < ln; ++i) for (int j = 0; j < rn; ++j)
        if (r[j].start - l[i].end - 1 >= p->min_len && tot > l[i].ed + r[j].ed) {
            tot = l[i].ed + r[j].ed; *ts = l[i].end + 1; *te = r[j].start - 1;
        }
    return tot;
}

/* src/gen_cons.c:128-171 single_copy_full_len_seq() */
static void single_copy(int seq_len, const char *seq, tho_read_res_t *out, const tho_para_t *p)
{
    int cons_len = 0, full_length = 0, ts = -1, te = -1, tot, pos[2] = { 0, 0 };
    ed_res_t r5[3], r3[3];
    int n5 = collect_ed(p, p->five_seq, p->five_len, seq, seq_len, r5);
    int n3 = collect_ed(p, p->three_rc_seq, p->three_len, seq, seq_len, r3);
    tot = full_len_pick(p, n5, r5, n3, r3, &ts, &te);
    if (tot != INT_MAX) { pos[0] = ts; pos[1] = te; cons_len = te - ts + 1; full_length = 1; }
    if (tot > 0) {
        n5 = collect_ed(p, p->five_rc_seq, p->five_len, seq, seq_len, r5);
        n3 = collect_ed(p, p->three_seq, p->three_len, seq, seq_len, r3);
        if (full_len_pick(p, n3, r3, n5, r5, &ts, &te) < tot) { pos[0] = ts; pos[1] = te; cons_len = te - ts + 1; full_length = 2; }
    }
    if (full_length > 0) {
        if (p->only_unit) write_unit(out, pos, 2);
        else {
            uint8_t *q = NULL;
            if (p->out_fmt == THO_FASTQ_FMT || p->out_fmt == THO_TAB_QUAL_FMT) { q = (uint8_t *)malloc((size_t)cons_len + 1); memset(q, 33, cons_len); }
            write_cons(out, seq + pos[0], q, cons_len, pos[0], pos[1], 1.0, 100.0, p, full_length, pos, 2);
            free(q);
        }
    }
}

/* src/tidehunter.c:23-60 tidehunter_core() */
int tho_core(const char *seq, int len, const tho_para_t *p, tho_read_res_t *out)
{
    if (len < p->k) return 0;
    uint8_t *bseq = (uint8_t *)malloc((size_t)len);
    tho_encode(seq, len, bseq);
    uint64_t *hits;
    int hit_n = tho_collect_hits(bseq, len, p, &hits);
    tho_chain_res_t cr;
    tho_chain(hits, hit_n, p, &cr);
    free(hits);
    for (int c = 0; c < cr.ch_n; ++c) {
        int par_n, *par_pos = tho_partition(bseq, len, &cr, &cr.chain[c], p, &par_n);
        if (par_n >= p->min_copy + 1) seqs_msa(len, bseq, par_n, par_pos, out, p);
        free(par_pos);
    }
    free(bseq);
    if (p->single_copy == 1 && p->only_full_length && p->five_seq && p->three_seq) single_copy(len, seq, out, p);
    tho_chain_free(&cr);
    return 0;
}
