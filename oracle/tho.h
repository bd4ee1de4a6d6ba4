/*
 * oracle/tho.h -- CPU restatement ("oracle") of TideHunter's per-read hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link, import or run it, and only as the checker / CPU baseline.
 * The product (tidehunter_amd/) never includes or links any of this.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Parity status:
 *   - stages 0-3 + record arithmetic + formatting: pinned by README.md:222
 *     (test_50x4 end-to-end line) and by the values SURVEY.md section 4 records
 *     for full_length.fa / test_1000x10.fa (measured there on the reference's
 *     own code).
 *   - ksw2 / edlib wrappers: pinned function-by-function against oracle/_ref
 *     (the reference's own ksw2/, src/ksw2_align.c, edlib/, src/edlib_align.c
 *     compiled where they lie; see oracle/Makefile).
 *   - POA (abPOA): PARITY UNPINNED.  abPOA is an un-vendored, empty git
 *     submodule of the reference (.gitmodules:1-4, pinned commit unknown);
 *     tho_poa.c restates the published abPOA algorithm (adaptive-band
 *     partial-order alignment, convex gaps, row-column consensus) and is the
 *     de-facto oracle for the HIP POA kernel.  It is cross-checked against
 *     tho_poa_exact.c, an independent full-matrix POA in Kahn order (the node
 *     order never changes a consensus; the adaptive band, part of the
 *     algorithm, rarely does: tests/test_poa_exact.py, tests/test_gpu_scale.py).
 *     Claims are "GPU == this CPU POA", never "== reference", whenever more
 *     than two noisy units reach POA.
 */
#ifndef THO_H
#define THO_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THO_FASTA_FMT 1
#define THO_TAB_FMT 2
#define THO_FASTQ_FMT 3
#define THO_TAB_QUAL_FMT 4

/* mirrors mini_tandem_para, src/tidehunter.h:47-61 (defaults :9-41) */
typedef struct {
    int k, w, hpc;
    int min_copy, min_cov;
    double max_div, min_frac;
    int64_t min_p, max_p;
    int match, mismatch, gap_open1, gap_open2, gap_ext1, gap_ext2;
    float ada_match_rat;
    char *five_seq, *five_rc_seq; int five_len;
    char *three_seq, *three_rc_seq; int three_len;
    int out_fmt, min_len, only_unit, only_longest, only_full_length, single_copy;
    int n_thread;
} tho_para_t;

void tho_para_init(tho_para_t *p);                 /* src/main.c:329-364 */
int  tho_para_load_adapters(tho_para_t *p, const char *five_fn, const char *three_fn); /* src/main.c:389-394 */
void tho_para_free_adapters(tho_para_t *p);

/* ---- stage 0: encoding (src/seq.c:15-32,77-95) ---- */
void  tho_encode(const char *seq, int len, uint8_t *bseq);
char *tho_rc_seq(const char *seq, int len);

/* ---- stage 1: self hits (src/tandem_hit.c) ---- */
/* returns number of k-mer entries written to h (key<<32|pos) */
int tho_kmer_entries(const uint8_t *bseq, int len, const tho_para_t *p, uint64_t *h);
/* returns hit_n; *hits malloc'd (end<<32|period), sorted ascending */
int tho_collect_hits(const uint8_t *bseq, int len, const tho_para_t *p, uint64_t **hits);

/* ---- stage 2: chaining (src/tandem_chain.c) ---- */
typedef struct { int i, j; } tho_cell_t;
typedef struct {
    int from_i, from_j, start, end, score; int8_t is_tracked;
} tho_dp_t;
typedef struct {
    tho_cell_t *cell; int len, score;
    int est_ch_i, est_period, est_start;
} tho_chain_t;
typedef struct {
    int tot_n;            /* number of dp rows */
    int *row_size;        /* cells per row */
    tho_dp_t **dp;        /* dp[row][cell] */
    int ch_n;             /* chains returned (ascending end) */
    tho_chain_t *chain;
} tho_chain_res_t;
void tho_chain(const uint64_t *hits, int hit_n, const tho_para_t *p, tho_chain_res_t *res);
void tho_chain_free(tho_chain_res_t *res);

/* ---- ksw2 restatement (ksw2/ksw2_extz2_sse.c, ksw2/ksw2.h, src/ksw2_align.c) ---- */
/* global alignment, left-aligned gaps; returns iden_n; cigar is len<<4|op (0=M 1=I 2=D), caller frees */
int  tho_ksw2_global_cigar(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int *n_cigar, uint32_t **cigar, int *score);
int  tho_ksw2_global(const uint8_t *q, int qlen, const uint8_t *t, int tlen);
int  tho_ksw2_backtrack_left_end(int n_cigar, const uint32_t *cigar, int qlen, int tlen, int q_left_ext);
void tho_ksw2_right_ext(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int *max_q, int *max_t);
void tho_ksw2_left_ext(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int *max_q, int *max_t);

/* ---- stage 3: partition (src/partition.c:171-276) ---- */
/* returns malloc'd par_pos (NULL if par_n==0) */
int *tho_partition(const uint8_t *bseq, int len, const tho_chain_res_t *cr, const tho_chain_t *ch, const tho_para_t *p, int *par_n);

/* ---- edlib HW/LOC restatement (src/edlib_align.c:73-85, edlib/src/edlib.cpp:141-275,526-683) ---- */
int tho_edlib_hw(const char *q, int qlen, const char *t, int tlen, int *start, int *end, int k);

/* ---- POA (own restatement of the published abPOA algorithm; PARITY UNPINNED) ---- */
/* aligns seqs[1..n) progressively onto a graph seeded with seqs[0]; returns cons_len;
 * cons (codes 0..4) and cov must hold sum(lens) entries */
int tho_poa_consensus(int n_seqs, const uint8_t *const *seqs, const int *lens, const tho_para_t *p, uint8_t *cons, int *cov);
/* named switches of the restated POA (process-wide; tests only) */
#define THO_POA_GAP_WINS_TIE 0     /* default of THO_POA_OPT_GAP_TIE: a base that is as frequent as the gap stays in the consensus */
#define THO_POA_OPT_GAP_TIE 1
#define THO_POA_OPT_BAND 2         /* 1 = adaptive band (default), 0 = full rows */
void tho_poa_set_opt(int what, int value);
/* independent exact POA (tho_poa_exact.c): full matrix, Kahn order recomputed per sequence; flags bit 0: rank consensus
 * columns breadth-first instead of depth-first, bit 1: gap wins column ties */
int tho_poa_exact(int n_seqs, const uint8_t *const *seqs, const int *lens, const tho_para_t *p, int flags, uint8_t *cons, int *cov);

/* ---- stage 4: consensus records (src/gen_cons.c, src/abpoa_cons.c) ---- */
typedef struct {
    char *seq; char *qual;      /* cons_len chars each (qual may be NULL) */
    int cons_len, start, end;   /* 0-based as stored in tandem_seq_t */
    double copy_num, ave_match;
    int full_length;
    int *sub_pos; int pos_n;
    /* integer evidence kept for parity tests against the HIP path */
    int n_seqs, max_q_l, max_t_l, max_q_r, max_t_r;
} tho_cons_t;
typedef struct { int n, m; tho_cons_t *a; } tho_read_res_t;
void tho_read_res_clear(tho_read_res_t *r);
void tho_read_res_free(tho_read_res_t *r);

/* per-read driver, src/tidehunter.c:23-60 */
int tho_core(const char *seq, int len, const tho_para_t *p, tho_read_res_t *out);

/* ---- I/O ---- */
typedef struct { char *name; char *seq; int len; } tho_read_t;
typedef struct tho_reader tho_reader_t;
tho_reader_t *tho_reader_open(const char *fn);
int  tho_reader_next(tho_reader_t *r, tho_read_t *out); /* 1 = got one, 0 = EOF; out fields malloc'd */
void tho_reader_close(tho_reader_t *r);
void tho_read_free(tho_read_t *r);
/* src/main.c:214-271 */
void tho_output(FILE *fp, const tho_read_t *rd, const tho_read_res_t *res, const tho_para_t *p);

/* whole-file driver with the reference's one-ticket-per-read pthread queue (src/main.c:273-291,375-436) */
int tho_run_file(const char *fn, const tho_para_t *p, FILE *out);

#ifdef __cplusplus
}
#endif
#endif
