/*
 * include/tidehunter_amd.h -- C-ABI of the MI355X-native TideHunter hot path.
 *
 * Plain C, plain pointers and sizes, no torch/HIP types.  Each entry point names the reference
 * interface it stands in for (paths relative to the TideHunter v1.5.5 tree).  The library is a
 * drop-in for the per-read hot path only: k-mer self-hits, chaining, partition, POA consensus and
 * the consensus statistics, i.e. what tidehunter_core() (src/tidehunter.c:23-60) does for every
 * read of a chunk inside mini_tandem() (src/main.c:402-425).  There is NO CPU fallback: every
 * compute entry point fails (NULL / negative return, message on stderr and in th_last_error())
 * when no AMD GPU / HIP runtime is present.
 *
 * Error convention: the reference terminates the process on any error (err_fatal,
 * src/utils.c:91-117).  A library cannot do that to its host, so functions return -1 (or NULL) after
 * printing "[tidehunter_amd] <message>" to stderr; partial output is never produced.  The bundled
 * command line (TideHunter-amd) turns that into exit status 1, like the reference.
 */
#ifndef TIDEHUNTER_AMD_H
#define TIDEHUNTER_AMD_H
#include <stdint.h>
#include <stdio.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TH_FASTA_FMT 1      /* src/tidehunter.h:36-39 */
#define TH_TAB_FMT 2
#define TH_FASTQ_FMT 3
#define TH_TAB_QUAL_FMT 4

/* mini_tandem_para, src/tidehunter.h:47-61: SAME fields in the SAME order and types, so the layouts are identical and a
 * maintainer may pass a mini_tandem_para* where a th_para_t* is expected (tests/test_boundary.py checks the field lists against
 * the reference header).  cons_out / detail_fp are carried but not used by the library: the caller owns the output stream. */
typedef struct {
    int64_t k, w; int hpc;
    int min_copy, min_cov; double max_div, div_exp, min_frac;
    int64_t min_p, max_p;
    int64_t max_range;
    int match, mismatch, gap_open1, gap_open2, gap_ext1, gap_ext2;
    float ada_match_rat;
    char *five_fn, *five_seq, *five_rc_seq; int five_len;
    char *three_fn, *three_seq, *three_rc_seq; int three_len;
    int out_fmt, min_len, only_unit, only_longest, only_full_length, single_copy;
    FILE *cons_out, *detail_fp;
    int n_thread;
} th_para_t;

/* kstring_t (src/kseq.h:84-87) and seq_t (src/seq.h:6-8), same layouts */
typedef struct { size_t l, m; char *s; } th_kstring_t;
typedef struct { th_kstring_t name, seq, comment, qual; } th_seq_t;

/* tandem_seq_t, src/tidehunter.h:63-72: identical layout.  The consensus strings of one read are concatenated in
 * cons_seq->seq.s (qualities in cons_seq->qual.s) and split by cons_len[], exactly as in the reference (src/main.c:253-263),
 * so the reference's own mini_tandem_output() can print slots filled by th_process_batch(). */
typedef struct {
    th_seq_t *cons_seq;
    int cons_n, cons_m;
    int *cons_start, *cons_end, *cons_len; double *copy_num, *ave_match;
    int8_t *full_length;
    int **sub_pos; int *pos_n, *pos_m;
    int *cons_score;
} th_tandem_seq_t;

typedef struct th_handle th_handle_t;

/* mini_tandem_init_para() / mini_tandem_free_para(), src/main.c:329-373 */
th_para_t *th_init_para(void);
void th_free_para(th_para_t *p);
/* adapter loading as in mini_tandem(), src/main.c:389-394 (needs five_fn and three_fn set) */
int th_load_adapters(th_para_t *p);

/* alloc_tandem_seq() / tandem_seq_free(), src/main.c:293-324 */
th_tandem_seq_t *th_alloc_tandem_seq(int n);
void th_free_tandem_seq(th_tandem_seq_t *t, int n);

/* aux_init() / aux_free(), src/main.c:184-206: one handle = one GPU (device ordinal) + its stream,
 * buffers and the read-only parameter snapshot.  Not re-entrant per handle; use one per thread/GPU. */
th_handle_t *th_init(const th_para_t *p, int device);
void th_destroy(th_handle_t *h);

/* The batch seam, src/main.c:402-425: fills out[0..n) so that th_output() prints what
 * mini_tandem_output() would.  Equivalent to calling tidehunter_core() (src/tidehunter.h:85) on every
 * read.  out[]: slots laid out as the reference's alloc_tandem_seq() (src/main.c:293-311) or th_alloc_tandem_seq()
 * allocate them (the same allocation: INTEGRATION.md passes the reference's own); slots are reset first. */
int th_process_batch(th_handle_t *h, const char *const *seqs, const int *lens, int n, th_tandem_seq_t *out);

/* The same in three steps, so that a caller (bench.py) can keep a batch resident in HBM and time
 * the device path alone:  upload (H2D) -> run (all kernels + small D2H) -> collect (host-side
 * double arithmetic, adapter search, filters; src/gen_cons.c:208-301, src/abpoa_cons.c:92-108). */
int th_batch_upload(th_handle_t *h, const char *const *seqs, const int *lens, int n);
/* optional: sizes the device buffers for batches of up to max_reads reads / max_bases bases, so that growing batches do not re-allocate */
int th_batch_reserve(th_handle_t *h, int max_reads, long long max_bases);
int th_batch_run(th_handle_t *h);
int th_batch_collect(th_handle_t *h, th_tandem_seq_t *out);
/* device milliseconds of the last th_batch_run() per kernel, measured with HIP events on the library's own
 * stream(s): [0] encode+hits [1] chain_dp [2] chain_extract [3] partition [4] poa [5] align; returns 6 */
int th_batch_timing(th_handle_t *h, double *ms_kernel, int n);
/* TH_SUBBATCH=2..4 cuts a batch of 16384 reads or more (TH_SUBBATCH_MIN) into that many parts that run concurrently on streams of
 * their own (round 2's default; one part since the host left the middle of the pipeline); th_batch_timing sums the parts,
 * th_batch_launches returns how many the last run had. */
int th_batch_launches(th_handle_t *h);
/* host synchronisations (stream drains) of the last th_batch_run(): three per batch -- after the job tables, after the POA, at the end */
int th_batch_syncs(th_handle_t *h);
/* cumulative counters of the handle since th_init(): out[0] consensus jobs left WITHOUT a consensus (a POA job the kernels could not serve even on their
 * slow paths -- abPOA has no such failure mode, src/abpoa_cons.c:86-87, so anything but 0 is output that differs from the reference's; a warning goes to stderr),
 * out[1] POA jobs given back and rerun (scratch estimate, int16 range, adaptive band), out[2] reruns of stages 2-3 / the job tables because an arena or a table
 * was too small (the reference allocates the worst case, src/tandem_chain.c:361-368), out[3] batches, out[4] consensus jobs; returns 5 */
int th_batch_stats(th_handle_t *h, long long *out, int n);

/* The batch seam over a SEQUENCE of batches, pipelined: while the kernels of batch i run, batch i+1 is uploaded (own copy stream,
 * second input set in HBM) and the record arithmetic of batch i-1 fills out[i-1][..] (second result set).  Same results as
 * n_batches calls of th_process_batch(); this is the loop body of mini_tandem() (src/main.c:396-432) with its three phases
 * overlapped.  th_run_file() drives the same pipeline from the reader. */
int th_process_batches(th_handle_t *h, int n_batches, const char *const *const *seqs, const int *const *lens, const int *n, th_tandem_seq_t *const *out);
/* the resident batch (th_batch_upload) run `steps` times, the record arithmetic of pass k under the kernels of pass k+1; ms_kernel
 * (8 doubles, may be NULL): device milliseconds per kernel summed over the passes -- bench.py's timed region */
int th_batch_run_steps(th_handle_t *h, int steps, th_tandem_seq_t *out, double *ms_kernel);

/* mini_tandem_output(), src/main.c:214-271 (resets the slots after printing, as the reference does) */
void th_output(FILE *fp, int n, const char *const *names, const char *const *seqs, const int *lens,
               th_tandem_seq_t *tseq, const th_para_t *p);

/* the same text into memory (malloc'ed, NUL-terminated, *len bytes; release with th_free_buffer): a rank of a one-process-per-GPU launch
 * formats its block and the blocks are gathered in rank order = input order (src/main.c:214-217) */
char *th_format_output(int n, const char *const *names, const char *const *seqs, const int *lens, th_tandem_seq_t *tseq, const th_para_t *p, size_t *len);
void th_free_buffer(void *p);

/* mini_tandem(), src/main.c:375-436: whole-file driver (FASTA/FASTQ, optionally gzipped) */
int th_run_file(const char *read_fn, th_para_t *p, FILE *out, int device, int batch_reads);
/* the same over several devices of one node: one worker per device takes the next batch from a shared ticket counter -- the
 * reference's THREAD_READ_I (src/main.c:273-291) at batch granularity, so a device that finishes early takes more batches
 * (reads are independent, src/tidehunter.c:23-60: the shards exchange nothing); output order = input order.  1 <= n_dev <= 16. */
int th_run_file_multi(const char *read_fn, th_para_t *p, FILE *out, const int *devices, int n_dev, int batch_reads);

/* The FASTA/FASTQ reader alone (kseq grammar, src/kseq.h:176-215; the reader of mini_tandem_read_seq, src/main.c:173-182), no device
 * involved: an uncompressed file is mmap'ed and parsed in place (single-line sequences are never copied), a bgzip file is inflated by
 * a pool of threads, plain gzip / stdin stream through zlib on a producer thread.  Returns the number of records, -1 if the file
 * cannot be opened, -2 if it is corrupt or truncated (th_run_file then fails too instead of returning a shortened output); *n_bases and *digest (FNV-1a over names and sequences) are optional. */
long long th_parse_file(const char *read_fn, long long *n_bases, unsigned long long *digest);

/* deterministic synthetic tandem-repeat reads; returns read length, writes at most cap bases */
int th_synth_read(uint64_t seed, uint64_t read_index, int period, int copies, double err, char *dst, int cap);

/* config 5 of SURVEY.md 8(d): adapter-bounded units of mixed periods (200..5000 by read_index mod 5), both strands; -1 if the
 * adapters do not fit the shortest period.  th_synth_adapter writes len bases + NUL (which: 0 = 5' adapter, 1 = 3' adapter). */
void th_synth_adapter(uint64_t seed, int which, char *dst, int len);
int th_synth_read_c5(uint64_t seed, uint64_t read_index, const char *five, int five_len, const char *three, int three_len, double err, char *dst, int cap);

/* Shared ticket counters for one-process-per-GPU launches (bench.py --steal): the multi-process form of the reference's ticket
 * counter, src/main.c:208-210,277-281.  `name` is a POSIX shared-memory name ("/th_job_1234"); one rank creates, all open;
 * th_ticket_next returns the previous value of counter `counter` (0..511) and advances it atomically. */
typedef struct th_ticket th_ticket_t;
th_ticket_t *th_ticket_open(const char *name, int create);
long long th_ticket_next(th_ticket_t *t, int counter);
void th_ticket_close(th_ticket_t *t, int unlink_it);

const char *th_last_error(void);
const char *th_version(void);

#ifdef __cplusplus
}
#endif
#endif
