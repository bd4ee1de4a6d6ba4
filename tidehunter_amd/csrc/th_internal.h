/* th_internal.h -- interface between the C host layer (th_host.c) and the device pipeline
 * (th_pipeline.hip).  Not part of the public C-ABI (include/tidehunter_amd.h). */
#ifndef TH_INTERNAL_H
#define TH_INTERNAL_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int k, w, hpc; uint32_t min_p, max_p; double one_minus_div; int min_copy;
    int match, mismatch, go1, ge1, go2, ge2;
    int only_unit;          /* -u: stop after partition (no POA, no alignment) */
    int want_cov;           /* per-base coverage is needed on the host (-r / -c filters, FASTQ / -f 4 qualities) */
    /* adapter search on the device (src/gen_cons.c:226-291): the four queries in the order 5', rc(3'), rc(5'), 3' with their
     * lengths and edlib thresholds k = (int)(len * (1 - ada_match_rat)) (float arithmetic, done by the host); NULL = no adapters */
    const char *ad_seq[4]; int ad_len[4], ad_k[4];
    int single_copy;        /* -s with -F and adapters: the four queries are also searched in every whole read (src/gen_cons.c:128-171) */
} th_gpu_para_t;

/* one consensus job as the host sees it after the device run */
typedef struct {
    int read;               /* read index in the batch */
    int pos_off, pos_n;     /* boundaries of the run inside par[] */
    int n_seqs;             /* units handed to POA (src/abpoa_cons.c:36-48) */
    int cons_off, cons_len; /* consensus codes / coverage inside cons[] / cov[] (cons_len 0 = none) */
    int aln_off;            /* iden[aln_off + u] = identical columns of unit u vs consensus */
    int status;             /* POA kernel status, 0 = ok */
} th_job_t;                 /* (written by the device: th_k_jobs_emit / th_k_items_emit) */

typedef struct {
    int n_jobs; const th_job_t *jobs;
    const int *par;         /* compact par_pos values */
    const uint8_t *cons; const int *cov; const int *iden;   /* cov is NULL unless want_cov */
    const int *ext;         /* 4 ints per job: max_q_l, max_t_l, max_q_r, max_t_r of the ksw2 extensions (src/gen_cons.c:218-223) */
    const int *single;      /* -s: 40 ints per read -- per query (5', rc(3'), rc(5'), 3') the hit count and up to three (ed, start, end), as collect_ed_res
                             * leaves them (src/gen_cons.c:89-110); NULL when the device did not search (no -s, or an adapter longer than 256 symbols) */
    const int *adapt;       /* 12 ints per job: (ed, start, end) of the four adapter searches on consensus + consensus; NULL when the
                             * device did not run them (no adapters, or an adapter longer than 256 symbols: the host searches) */
    double ms_kernels;      /* device time of the last run (HIP events), all kernels */
    double ms_dominant;     /* device time of the slowest kernel of the last run */
    double ms_kernel[8];    /* per kernel: 0 encode+hits, 1 chain_dp, 2 chain_extract, 3 partition, 4 poa, 5 align */
} th_gpu_out_t;

typedef struct th_gpu th_gpu_t;

th_gpu_t *th_gpu_create(const th_gpu_para_t *p, int device);
void th_gpu_destroy(th_gpu_t *g);
/* sizes the per-read / per-base buffers for batches of up to n reads and T bases (optional; avoids re-allocation when batches grow) */
int th_gpu_reserve(th_gpu_t *g, int n, long long T);
int th_gpu_room_for(th_gpu_t *g, int n, long long T);     /* 1 if a further pipeline of this batch size fits the device memory */
/* A pipeline keeps TWO input sets and TWO result sets (slot 0 / 1), so that a driver can upload batch i+1 and do the record
 * arithmetic of batch i-1 while the kernels of batch i run: upload(slot) may overlap run() of the other slot (own copy stream),
 * get(slot) stays valid until the next run into that slot.  One thread at a time per function; run() is the only one that
 * touches the work buffers. */
/* copies the reads through pinned staging buffers to HBM; wait != 0: returns when the batch is resident */
int th_gpu_upload(th_gpu_t *g, int slot, const char *const *seqs, const int *lens, int n, int wait);
/* runs the whole device path on input set `in_slot` and brings the small result records back into result set `out_slot` */
int th_gpu_run(th_gpu_t *g, int in_slot, int out_slot);
int th_gpu_get(th_gpu_t *g, int out_slot, th_gpu_out_t *out);
/* host synchronisations (hipStreamSynchronize / event waits) of the last run, and seconds this handle spent in device malloc so far */
int th_gpu_syncs(th_gpu_t *g);
/* cumulative counters of the handle's recovery paths: [0] consensus jobs left WITHOUT a consensus (a POA job the kernels could not serve: output that differs
 * from the reference's, which has no such failure mode), [1] POA jobs given back and rerun, [2] reruns of stages 2-3 / the job tables because an arena or a
 * table was too small, [3] batches, [4] consensus jobs; returns 5 */
int th_gpu_stats(th_gpu_t *g, long long *out, int n);
const char *th_gpu_error(void);

/* test hook: copies an intermediate stage of the resident batch to the host.
 * what: 0 bseq(u8) 1 hit_n(i32/read) 2 hit_end 3 hit_per 4 score 5 from (i32, read-major at roff)
 *       6 chain records (8 x i32 each, grouped by read in emission order) 7 cells (i32)
 *       8 roff (u32, n+1)   returns bytes written or -1 */
long long th_gpu_debug(th_gpu_t *g, int what, void *dst, long long cap);
/* the same through a handle (exported for tests/, not declared in the public header) */
struct th_handle;
long long th_debug_stage(struct th_handle *h, int what, void *dst, long long cap);


/* test hooks (th_debug_hooks.h): RAW pairs through the device's alignment / adapter engines, for the reference-generated vectors of
 * tests/golden/ (tests/test_gpu_ref_vectors.py).  Sequences are concatenated, x_off has n + 1 entries.
 * th_debug_align_pairs: kind 0 ksw2_global -> out[i] = iden_n; kind 1 / 2 ksw2_left_ext / ksw2_right_ext -> out[2i], out[2i+1] = (max_q, max_t), all through
 * th_k_align; cfg = { variant 0 <12288,2> / 1 <16384,6> / 2 <24576,6>, TH_ALN_MODE, wdiv, wadd, wtol, xpct, defer }; stats (8 ints, may be NULL):
 * error flags, deferred items, wide first bands tried / failed.
 * th_debug_global_pairs: ksw2_global_with_cigar + ksw2_backtrack_left_end on the partition's engines (banded 0: the full matrix; 1: its flow of round 4 --
 * certified band on the packed strip engine, pruned wider band, full matrix).
 * th_debug_adapter: edlib_align_HW of four queries against every target (doubled 0: raw bytes, th_k_adapter_reads; 1: codes, searched over target + target, th_k_adapter). */
int th_debug_align_pairs(int device, int kind, int n, const uint8_t *q_cat, const int *q_off, const uint8_t *t_cat, const int *t_off, const int *cfg, int *out, int *stats);
int th_debug_global_pairs(int device, int n, const uint8_t *q_cat, const int *q_off, const uint8_t *t_cat, const int *t_off, const int *q_left_ext, int n_ext, int *iden, int *tback, int banded);
int th_debug_adapter(int device, int doubled, const char *const queries[4], const int qlen[4], const int k[4], int n, const uint8_t *t_cat, const int *t_off, int *out);

/* host CPU budget (th_host.c; exported for bench.py and the tests, not part of the public header) */
int th_effective_cpus(void);                       /* CPUs the process may really use: affinity mask cut down to the cgroup quota */
int th_host_cpu_seconds(double *out, int n);       /* CPU seconds so far by role: [0] reader [1] inflate [2] upload [3] run [4] record arithmetic + formatting [5] writer; returns 6 */

#ifdef __cplusplus
}
#endif
#endif
