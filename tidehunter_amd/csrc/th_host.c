/*
 * th_host.c -- C host layer of the MI355X TideHunter hot path: parameters, the batch seam, the
 * floating-point record arithmetic the reference keeps on the host, adapter search, output
 * formatting and the whole-file driver.  All device work is behind th_internal.h.
 *
 * Mirrors (same names, argument meaning, error behaviour where a library can):
 *   mini_tandem_init_para / mini_tandem                      src/main.c:329-364, :375-436
 *   seqs_msa (host part) / write_tandem_cons_seq / _unit      src/gen_cons.c:10-83, :173-301
 *   abpoa_gen_cons (host part: min-cov filter, Phred)         src/abpoa_cons.c:49-108
 *   single_copy_full_len_seq / collect_ed_res                 src/gen_cons.c:89-171
 *   edlib_align_HW                                            src/edlib_align.c:73-85
 *   mini_tandem_output                                        src/main.c:214-271
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <sched.h>
#include <sys/resource.h>
#include <limits.h>
#include <ctype.h>
#include <zlib.h>
#include <pthread.h>
#include <unistd.h>
#include <time.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <stddef.h>
#include "tidehunter_amd.h"
#include "th_internal.h"

static __thread char g_err[512];   /* per thread, like the device layer's message */
const char *th_last_error(void) { return g_err[0] ? g_err : th_gpu_error(); }
const char *th_version(void) { return "1.5.5-mi355x.r3"; }
static int host_fail(const char *m) { snprintf(g_err, sizeof(g_err), "%s", m); fprintf(stderr, "[tidehunter_amd] %s\n", m); return -1; }

/* A handle owns up to two device pipelines (own stream, own buffers) on its GPU.  A large batch is cut in two halves balanced by
 * bases that run CONCURRENTLY: the three large kernels each leave vector-issue slots idle (0.75-0.96 busy) and every pipeline has host
 * gaps between its launches, and a second stream fills both -- measured with two ranks sharing one GPU: +7 % reads/s at the headline
 * shape, +20 % on config 5 (tools/gpu_tworank.sh).  Reads are independent, so the halves exchange nothing; results are merged by
 * read index.  Small batches (tests, the file driver's own two workers) use one pipeline. */
#define TH_MAX_SUB 4
typedef struct { int n, n_sub, sub_off[TH_MAX_SUB + 1]; const char *const *seqs; const int *lens; } th_batch_t;   /* one batch (reads borrowed until its collect) */
struct th_handle {
    th_para_t para;              /* snapshot (adapter strings are borrowed from the caller's th_para_t) */
    th_gpu_t *gpu[TH_MAX_SUB]; int n_sub_max, sub_min;
    /* Two input sets and two result sets per pipeline (th_internal.h): `up[s]` is the batch uploaded into input set s, `res[s]` the
     * batch whose results sit in result set s.  The three-step calls of the public API use set 0; the file driver and
     * th_process_batches() alternate, so that the upload of batch i+1 and the record arithmetic of batch i-1 overlap the kernels of batch i. */
    th_batch_t up[2], res[2];
};

/* ---------------------------------------------------------------- parameters */
th_para_t *th_init_para(void)
{
    th_para_t *p = (th_para_t *)calloc(1, sizeof(th_para_t));
    p->n_thread = 4; p->ada_match_rat = 0.8f;
    p->k = 8; p->w = 1; p->hpc = 0; p->min_copy = 2; p->max_div = 0.25; p->div_exp = exp(2 * 8 * 0.25);
    p->min_p = 30; p->max_p = 10000; p->min_cov = 0; p->min_frac = 0.0;
    p->match = 2; p->mismatch = 4; p->gap_open1 = 4; p->gap_open2 = 24; p->gap_ext1 = 2; p->gap_ext2 = 1;
    p->min_len = 30; p->out_fmt = TH_FASTA_FMT; p->max_range = -1;
    return p;
}
void th_free_para(th_para_t *p)
{
    if (!p) return;
    free(p->five_fn); free(p->five_seq); free(p->five_rc_seq); free(p->three_fn); free(p->three_seq); free(p->three_rc_seq);
    free(p);
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* ---------------------------------------------------------------- host CPU budget */
/* CPU seconds by role, summed over every thread the host layer starts (CLOCK_THREAD_CPUTIME_ID, added when a thread ends or hands a
 * batch on); what the process spent beyond them -- getrusage minus the sum -- is the HIP runtime's own threads.  th_run_file prints the
 * table under TH_TRACE and bench.py reports it as host_cpu_s_per_1M_reads: the seam this layer replaces runs n_thread workers and one
 * writer (src/main.c:273-291, :214-271), and a node has perhaps two cores per GPU to give. */
enum { CPU_READER = 0, CPU_INFLATE, CPU_UPLOAD, CPU_RUN, CPU_COLLECT, CPU_WRITER, CPU_N };
static double g_cpu_s[CPU_N];
static pthread_mutex_t g_cpu_mu = PTHREAD_MUTEX_INITIALIZER;
static double thread_cpu_s(void) { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void cpu_account(int cat, double s) { pthread_mutex_lock(&g_cpu_mu); g_cpu_s[cat] += s; pthread_mutex_unlock(&g_cpu_mu); }
int th_host_cpu_seconds(double *out, int n)       /* [0] reader [1] inflate [2] upload [3] run [4] collect + format [5] writer; returns 6 */
{
    pthread_mutex_lock(&g_cpu_mu);
    for (int i = 0; i < n && i < CPU_N; ++i) out[i] = g_cpu_s[i];
    pthread_mutex_unlock(&g_cpu_mu);
    return CPU_N;
}

/* CPUs this process may actually use: the affinity mask cut down to the cgroup's CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us /
 * cpu.cfs_period_us; the tightest limit on the way up from the process's own group).  sysconf(_SC_NPROCESSORS_ONLN) says 256 on a box
 * whose container may use 16. */
static double quota_of(const char *dir)
{
    char fn[600], buf[128]; FILE *f; double q = 0;
    snprintf(fn, sizeof(fn), "%s/cpu.max", dir);
    if ((f = fopen(fn, "r")) != NULL) {
        if (fgets(buf, sizeof(buf), f)) { double a, b; if (sscanf(buf, "%lf %lf", &a, &b) == 2 && b > 0) q = a / b; }
        fclose(f);
        return q;
    }
    snprintf(fn, sizeof(fn), "%s/cpu.cfs_quota_us", dir);
    if ((f = fopen(fn, "r")) != NULL) {
        double a = -1, b = 0; if (fscanf(f, "%lf", &a) != 1) a = -1;
        fclose(f);
        snprintf(fn, sizeof(fn), "%s/cpu.cfs_period_us", dir);
        if (a > 0 && (f = fopen(fn, "r")) != NULL) { if (fscanf(f, "%lf", &b) == 1 && b > 0) q = a / b; fclose(f); }
    }
    return q;
}
static int g_eff_cpus = 1;
static pthread_once_t g_eff_cpus_once = PTHREAD_ONCE_INIT;
static void effective_cpus_init(void)        /* (once: the reader thread and the collector threads may ask first at the same time) */
{
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) { long c = sysconf(_SC_NPROCESSORS_ONLN); n = c > 0 ? (int)c : 1; }
    double best = 0;
    /* /proc/self/cgroup: "0::<path>" (v2) and / or "<id>:<controllers>:<path>" (v1; the line whose controller list holds "cpu") */
    char rel2[400] = "", rel1[400] = "";
    FILE *f = fopen("/proc/self/cgroup", "r");
    if (f) {
        char line[512];
        while (fgets(line, sizeof(line), f)) {
            line[strcspn(line, "\n")] = 0;
            char *c1 = strchr(line, ':'), *c2 = c1 ? strchr(c1 + 1, ':') : NULL;
            if (!c2) continue;
            *c2 = 0;
            if (c1[1] == 0) { snprintf(rel2, sizeof(rel2), "%s", c2 + 1); continue; }
            char *save = NULL;
            for (char *tok = strtok_r(c1 + 1, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) if (!strcmp(tok, "cpu")) snprintf(rel1, sizeof(rel1), "%s", c2 + 1);
        }
        fclose(f);
    }
    const char *roots[2] = { "/sys/fs/cgroup", "/sys/fs/cgroup/cpu" }; const char *rels[2] = { rel2, rel1 };
    for (int r = 0; r < 2; ++r) {      /* the tightest limit on the way up from the process's own group to the root of the hierarchy */
        char dir[1100]; snprintf(dir, sizeof(dir), "%s%s", roots[r], rels[r]);
        size_t l = strlen(dir); while (l > 1 && dir[l - 1] == '/') dir[--l] = 0;
        for (;;) {
            const double q = quota_of(dir);
            if (q > 0 && (best == 0 || q < best)) best = q;
            if (strlen(dir) <= strlen(roots[r])) break;
            char *sl = strrchr(dir, '/');
            if (!sl) break;
            *sl = 0;
        }
    }
    if (best > 0) { const int q = (int)(best + 0.999); if (q < n) n = q; }
    { const char *e = getenv("TH_EFFECTIVE_CPUS"); if (e && atoi(e) > 0) n = atoi(e); }      /* (tests) */
    g_eff_cpus = n < 1 ? 1 : n;
}
int th_effective_cpus(void) { pthread_once(&g_eff_cpus_once, effective_cpus_init); return g_eff_cpus; }
static int g_live_handles = 0;      /* handles of this process: they share the CPUs */

/* ---------------------------------------------------------------- FASTA/FASTQ(.gz) reader (kseq grammar, src/kseq.h:176-215) */
/* Three sources behind one record grammar (the reference reads everything through one gzread loop, src/kseq.h:52-72):
 *   MEM     an uncompressed regular file is mmap'ed and parsed in place: a sequence that sits on one line is handed on as a
 *           pointer into the mapping (no copy at all -- the only copy left is the one into the pinned upload staging area, made
 *           by each device worker), only names and multi-line sequences are copied.  ~10 GB/s on one thread.
 *   BGZF    a block-gzipped file (bgzip: gzip members of <= 64 KiB whose compressed size sits in a 'BC' extra field) is mmap'ed,
 *           its members are inflated by a pool of threads straight to their final offsets, a group of members at a time, one
 *           group ahead of the parser.
 *   STREAM  anything else (plain gzip, stdin): zlib on a producer thread, double-buffered, so that inflation overlaps parsing
 *           (a single deflate stream cannot be inflated in parallel; ~0.4 GB/s). */
typedef struct { char *s; size_t l, m; } th_str_t;
static void str_putc(th_str_t *s, int c)
{
    if (s->l + 2 > s->m) { s->m = s->m ? s->m * 2 : 256; s->s = (char *)realloc(s->s, s->m); }
    s->s[s->l++] = (char)c; s->s[s->l] = 0;
}
static void str_append(th_str_t *s, const char *p, size_t n)
{
    if (s->l + n + 2 > s->m) { while (s->l + n + 2 > s->m) s->m = s->m ? s->m * 2 : 256; s->s = (char *)realloc(s->s, s->m); }
    memcpy(s->s + s->l, p, n); s->l += n; s->s[s->l] = 0;
}

enum { RD_STREAM = 0, RD_MEM = 1, RD_BGZF = 2 };
#define RD_CHUNK (4u << 20)
#define BGZF_GROUP 512                    /* members per group: <= 32 MiB of text */
typedef struct { const unsigned char *src; uint32_t clen, isize; size_t out_off; } bgzf_blk_t;
typedef struct th_reader {
    int mode, last, eof;
    int failed;                  /* corrupt or truncated input: what was parsed so far is NOT the whole file (th_run_file reports it) */
    /* the window the grammar reads from */
    const unsigned char *buf; size_t beg, end;
    /* STREAM: zlib on a producer thread, two buffers */
    gzFile fp; unsigned char *sbuf[2]; long slen[2]; int sfull[2], scur, sdone, squit; pthread_t sthr; pthread_mutex_t smu; pthread_cond_t scv; int sthr_on;
    /* MEM / BGZF: the mapping */
    const unsigned char *map; size_t map_len;
    /* MEM: the line index (below) */
    struct th_lidx *lx;
    /* BGZF: groups of members, inflated by a pool */
    size_t zpos; unsigned char *gbuf[2]; size_t gcap[2], glen[2]; int gcur, gpending, zbusy, zfinished; bgzf_blk_t *blk; int nblk; volatile int next_blk; int zerr;
    pthread_t zthr[16]; int nz; pthread_mutex_t zmu; pthread_cond_t zcv; int zgo, zdone, zquit;
} th_reader_t;

static void *stream_producer(void *arg)
{
    th_reader_t *r = (th_reader_t *)arg;
    for (int b = 0;; b ^= 1) {
        pthread_mutex_lock(&r->smu);
        while (r->sfull[b] && !r->squit) pthread_cond_wait(&r->scv, &r->smu);
        const int quit = r->squit;
        pthread_mutex_unlock(&r->smu);
        if (quit) { cpu_account(CPU_INFLATE, thread_cpu_s()); return NULL; }
        const long n = gzread(r->fp, r->sbuf[b], RD_CHUNK);
        pthread_mutex_lock(&r->smu);
        if (n < 0) { int en = 0; fprintf(stderr, "[tidehunter_amd] read error: %s\n", gzerror(r->fp, &en)); r->failed = 1; }
        r->slen[b] = n > 0 ? n : 0; r->sfull[b] = 1; if (n <= 0) r->sdone = 1;
        pthread_cond_broadcast(&r->scv); pthread_mutex_unlock(&r->smu);
        if (n <= 0) { cpu_account(CPU_INFLATE, thread_cpu_s()); return NULL; }
    }
}
/* BGZF member at p: returns its total size and the sizes of its deflate data / text, 0 if p is not a BGZF member */
static size_t bgzf_member(const unsigned char *p, size_t avail, uint32_t *clen, uint32_t *isize, const unsigned char **cdata)
{
    if (avail < 28 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const unsigned xlen = p[10] | p[11] << 8;
    if (12 + (size_t)xlen > avail) return 0;
    unsigned bsize = 0, off = 12;
    while (off + 4 <= 12 + xlen) {
        const unsigned slen = p[off + 2] | p[off + 3] << 8;
        if (p[off] == 'B' && p[off + 1] == 'C' && slen == 2 && off + 6 <= 12 + xlen) bsize = (p[off + 4] | p[off + 5] << 8) + 1u;
        off += 4 + slen;
    }
    if (!bsize || bsize > avail || bsize < 12 + xlen + 8) return 0;
    *cdata = p + 12 + xlen; *clen = bsize - 12 - xlen - 8;
    *isize = p[bsize - 4] | p[bsize - 3] << 8 | p[bsize - 2] << 16 | (uint32_t)p[bsize - 1] << 24;
    return bsize;
}
static void *bgzf_worker(void *arg)
{
    th_reader_t *r = (th_reader_t *)arg;
    z_stream z; int gen = 0;
    for (;;) {
        pthread_mutex_lock(&r->zmu);
        while (r->zgo == gen && !r->zquit) pthread_cond_wait(&r->zcv, &r->zmu);
        gen = r->zgo;
        const int quit = r->zquit;
        pthread_mutex_unlock(&r->zmu);
        if (quit) { cpu_account(CPU_INFLATE, thread_cpu_s()); return NULL; }
        unsigned char *out = r->gbuf[r->gpending];
        for (;;) {
            const int i = __atomic_fetch_add(&r->next_blk, 1, __ATOMIC_RELAXED);
            if (i >= r->nblk) break;
            const bgzf_blk_t *b = &r->blk[i];
            if (b->isize == 0) continue;                 /* the empty end-of-file member */
            memset(&z, 0, sizeof(z));
            if (inflateInit2(&z, -15) != Z_OK) { r->zerr = 1; continue; }
            z.next_in = (unsigned char *)b->src; z.avail_in = b->clen; z.next_out = out + b->out_off; z.avail_out = b->isize;
            const int rc = inflate(&z, Z_FINISH);
            if (rc != Z_STREAM_END || z.total_out != b->isize) r->zerr = 1;
            inflateEnd(&z);
        }
        pthread_mutex_lock(&r->zmu);
        if (++r->zdone == r->nz) pthread_cond_broadcast(&r->zcv);
        pthread_mutex_unlock(&r->zmu);
    }
}
/* indexes the next group of members and starts its inflation into gbuf[which]; returns 0 at the end of the file */
static int bgzf_start_group(th_reader_t *r, int which)
{
    size_t out = 0; int n = 0;
    while (n < BGZF_GROUP && r->zpos < r->map_len) {
        uint32_t clen, isize; const unsigned char *cd;
        const size_t sz = bgzf_member(r->map + r->zpos, r->map_len - r->zpos, &clen, &isize, &cd);
        if (!sz) { r->zerr = 1; break; }
        r->blk[n].src = cd; r->blk[n].clen = clen; r->blk[n].isize = isize; r->blk[n].out_off = out;
        out += isize; r->zpos += sz; ++n;
    }
    if (n == 0) return 0;
    if (out > r->gcap[which] || !r->gbuf[which]) { free(r->gbuf[which]); r->gcap[which] = out + (out >> 3) + 65536; r->gbuf[which] = (unsigned char *)malloc(r->gcap[which]); }    /* (never NULL: a group may hold only the empty end-of-file member, and inflate() rejects a NULL output) */
    r->glen[which] = out; r->nblk = n; r->next_blk = 0; r->gpending = which; r->zbusy = 1;
    pthread_mutex_lock(&r->zmu); r->zdone = 0; ++r->zgo; pthread_cond_broadcast(&r->zcv); pthread_mutex_unlock(&r->zmu);
    return 1;
}
static void bgzf_wait_group(th_reader_t *r)
{
    if (!r->zbusy) return;
    pthread_mutex_lock(&r->zmu);
    while (r->zdone < r->nz) pthread_cond_wait(&r->zcv, &r->zmu);
    pthread_mutex_unlock(&r->zmu);
    r->zbusy = 0;
}
/* ---- MEM mode: the line index.  Finding the ends of lines is all the grammar needs from the bytes of a sequence, and it is the one
 * part of parsing that costs per BYTE (memchr over 10 GB per million headline reads: ~1 s on the one thread that hands batches to every
 * device -- exactly the million reads per second eight GPUs take).  So helper threads scan the mapping ahead of the parser, a chunk at a
 * time, and leave the newline offsets of each chunk in an array; the parser walks those arrays and touches only the header lines.  The
 * split is by BYTES, not by records, so it needs no guess about where a record starts (a FASTQ quality line may begin with '@'): the
 * grammar still runs on one thread, in order, at a cost per LINE.  A chunk with more than one newline per 16 bytes is not indexed
 * (the parser falls back to memchr there), which bounds the index at a quarter of the text. */
#define LX_CHUNK ((size_t)4 << 20)
#define LX_AHEAD 48                      /* chunks the helpers may run ahead of the parser (192 MiB of page cache touched early) */
typedef struct { uint32_t *off; int n; int state; } lx_chunk_t;      /* state: 0 untouched, 1 being scanned, 2 ready (n < 0: dense, not indexed) */
typedef struct th_lidx {
    const unsigned char *map; size_t len; size_t n_chunk;
    lx_chunk_t *ch;
    pthread_mutex_t mu; pthread_cond_t cv;
    size_t next, parser_chunk; int quit;
    pthread_t thr[8]; int n_thr;
    size_t cur_chunk; int cur_i;          /* the parser's cursor: first newline not yet passed */
} th_lidx_t;
static void lx_scan(th_lidx_t *x, size_t c)
{
    const size_t a = c * LX_CHUNK, b = a + LX_CHUNK < x->len ? a + LX_CHUNK : x->len;
    const size_t cap_max = (b - a) / 16 + 16;
    size_t cap = 1024, n = 0;
    uint32_t *off = (uint32_t *)malloc(cap * sizeof(uint32_t));      /* (no memory: the chunk is marked dense below and the parser's own memchr takes it) */
    const unsigned char *p = x->map + a, *e = x->map + b;
    while (off && p < e) {
        const unsigned char *q = (const unsigned char *)memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        if (n == cap) {
            if (cap >= cap_max) { free(off); off = NULL; break; }
            cap = cap * 2 < cap_max ? cap * 2 : cap_max;
            uint32_t *grown = (uint32_t *)realloc(off, cap * sizeof(uint32_t));
            if (!grown) { free(off); off = NULL; break; }
            off = grown;
        }
        off[n++] = (uint32_t)(q - (x->map + a));
        p = q + 1;
    }
    pthread_mutex_lock(&x->mu);
    x->ch[c].off = off; x->ch[c].n = off ? (int)n : -1; x->ch[c].state = 2;
    pthread_cond_broadcast(&x->cv);
    pthread_mutex_unlock(&x->mu);
}
static void *lx_helper(void *arg)
{
    th_lidx_t *x = (th_lidx_t *)arg;
    for (;;) {
        pthread_mutex_lock(&x->mu);
        size_t c;
        for (;;) {
            if (x->quit) { pthread_mutex_unlock(&x->mu); cpu_account(CPU_READER, thread_cpu_s()); return NULL; }
            while (x->next < x->n_chunk && x->ch[x->next].state != 0) ++x->next;      /* (the parser may have taken one itself) */
            if (x->next >= x->n_chunk) { pthread_mutex_unlock(&x->mu); cpu_account(CPU_READER, thread_cpu_s()); return NULL; }
            if (x->next <= x->parser_chunk + LX_AHEAD) { c = x->next++; x->ch[c].state = 1; break; }
            pthread_cond_wait(&x->cv, &x->mu);
        }
        pthread_mutex_unlock(&x->mu);
        lx_scan(x, c);
    }
}
static th_lidx_t *lx_open(const unsigned char *map, size_t len)
{
    if (len < 4 * LX_CHUNK || getenv("TH_READER_NOINDEX")) return NULL;          /* small inputs: the parser's own memchr is cheaper than threads */
    th_lidx_t *x = (th_lidx_t *)calloc(1, sizeof(*x));
    if (!x) return NULL;                  /* (no index: the parser scans for newlines itself) */
    x->map = map; x->len = len; x->n_chunk = (len + LX_CHUNK - 1) / LX_CHUNK;
    x->ch = (lx_chunk_t *)calloc(x->n_chunk, sizeof(lx_chunk_t));
    if (!x->ch) { free(x); return NULL; }
    pthread_mutex_init(&x->mu, NULL); pthread_cond_init(&x->cv, NULL);
    int want = th_effective_cpus() >= 8 ? 3 : th_effective_cpus() >= 4 ? 2 : 1;
    { const char *e = getenv("TH_READER_THREADS"); if (e && atoi(e) >= 1 && atoi(e) <= 8) want = atoi(e); }
    for (int t = 0; t < want; ++t) if (pthread_create(&x->thr[x->n_thr], NULL, lx_helper, x) == 0) ++x->n_thr;
    return x;                              /* (no helper at all: the parser scans every chunk itself, below) */
}
static void lx_close(th_lidx_t *x)
{
    if (!x) return;
    pthread_mutex_lock(&x->mu); x->quit = 1; pthread_cond_broadcast(&x->cv); pthread_mutex_unlock(&x->mu);
    for (int t = 0; t < x->n_thr; ++t) pthread_join(x->thr[t], NULL);
    for (size_t c = 0; c < x->n_chunk; ++c) free(x->ch[c].off);
    free(x->ch); pthread_mutex_destroy(&x->mu); pthread_cond_destroy(&x->cv); free(x);
}
/* chunk c, ready: scanned here if no helper has taken it yet */
static lx_chunk_t *lx_get(th_lidx_t *x, size_t c)
{
    pthread_mutex_lock(&x->mu);
    if (c > x->parser_chunk) { x->parser_chunk = c; pthread_cond_broadcast(&x->cv); }
    if (x->ch[c].state == 0) { x->ch[c].state = 1; pthread_mutex_unlock(&x->mu); lx_scan(x, c); return &x->ch[c]; }
    while (x->ch[c].state != 2) pthread_cond_wait(&x->cv, &x->mu);
    pthread_mutex_unlock(&x->mu);
    return &x->ch[c];
}
/* first newline at or after byte `pos`, or the end of the mapping */
static size_t lx_next_nl(th_lidx_t *x, size_t pos)
{
    size_t c = pos / LX_CHUNK;
    if (c != x->cur_chunk) {
        for (size_t d = x->cur_chunk; d < c && d < x->n_chunk; ++d) { free(x->ch[d].off); x->ch[d].off = NULL; }      /* (passed for good: positions only grow) */
        x->cur_chunk = c; x->cur_i = 0;
    }
    for (; c < x->n_chunk; ++c) {
        lx_chunk_t *k = lx_get(x, c);
        const size_t base = c * LX_CHUNK;
        if (c != x->cur_chunk) { free(x->ch[x->cur_chunk].off); x->ch[x->cur_chunk].off = NULL; x->cur_chunk = c; x->cur_i = 0; }
        if (k->n < 0) {                    /* dense chunk: not indexed */
            const size_t from = pos > base ? pos : base, to = base + LX_CHUNK < x->len ? base + LX_CHUNK : x->len;
            const unsigned char *q = (const unsigned char *)memchr(x->map + from, '\n', to - from);
            if (q) return (size_t)(q - x->map);
            continue;
        }
        const uint32_t rel = pos > base ? (uint32_t)(pos - base) : 0;
        int i = x->cur_i;
        while (i < k->n && k->off[i] < rel) ++i;
        x->cur_i = i;
        if (i < k->n) return base + k->off[i];
    }
    return x->len;
}

/* next window of text; 0 at the end of the input */
static int rd_fill(th_reader_t *r)
{
    if (r->eof) return 0;
    if (r->mode == RD_MEM) { r->eof = 1; return 0; }                   /* the mapping was the only window */
    if (r->mode == RD_STREAM) {
        if (r->scur >= 0) {                                             /* hand the consumed buffer back */
            pthread_mutex_lock(&r->smu); r->sfull[r->scur] = 0; pthread_cond_broadcast(&r->scv); pthread_mutex_unlock(&r->smu);
        }
        const int b = r->scur < 0 ? 0 : r->scur ^ 1;
        pthread_mutex_lock(&r->smu);
        while (!r->sfull[b]) pthread_cond_wait(&r->scv, &r->smu);
        pthread_mutex_unlock(&r->smu);
        r->scur = b;
        if (r->slen[b] <= 0) { r->eof = 1; return 0; }
        r->buf = r->sbuf[b]; r->beg = 0; r->end = (size_t)r->slen[b];
        return 1;
    }
    /* BGZF: the group started earlier is (being) inflated; wait for it, expose it, start the one after it */
    for (;;) {
        if (r->zfinished) { if (r->zerr) { fprintf(stderr, "[tidehunter_amd] corrupt or truncated BGZF input\n"); r->failed = 1; } r->eof = 1; return 0; }
        if (!r->zbusy && !bgzf_start_group(r, 0)) { if (r->zerr) r->failed = 1; r->eof = 1; return 0; }
        bgzf_wait_group(r);
        if (r->zerr) { fprintf(stderr, "[tidehunter_amd] corrupt BGZF input\n"); r->failed = 1; r->eof = 1; return 0; }
        const int ready = r->gpending;
        r->buf = r->gbuf[ready]; r->beg = 0; r->end = r->glen[ready]; r->gcur = ready;
        if (!bgzf_start_group(r, ready ^ 1)) r->zfinished = 1;
        if (r->end > 0) return 1;
    }
}
static int rd_getc(th_reader_t *r)
{
    if (r->beg >= r->end && !rd_fill(r)) return -1;
    return r->buf[r->beg++];
}
/* Consumes the rest of the current line.  Appends it to s when s is not NULL; when zc is not NULL and the line lies in one
 * window, *zc / *zn receive a pointer into the window instead (MEM mode: it stays valid for the whole run).  Returns the byte
 * count (without the newline), -1 at the end of the input. */
static long rd_line(th_reader_t *r, th_str_t *s, const char **zc, size_t *zn)
{
    long got = 0;
    if (zc) *zc = NULL;
    for (;;) {
        if (r->beg >= r->end && !rd_fill(r)) return got ? got : -1;
        const unsigned char *p = r->buf + r->beg;
        const size_t avail = r->end - r->beg;
        const unsigned char *nl;
        if (r->lx) { const size_t at = lx_next_nl(r->lx, r->beg); nl = at < r->end ? r->buf + at : NULL; }      /* (MEM: buf is the mapping) */
        else nl = (const unsigned char *)memchr(p, '\n', avail);
        const size_t n = nl ? (size_t)(nl - p) : avail;
        if (zc && got == 0 && r->mode == RD_MEM) { *zc = (const char *)p; *zn = n; }        /* one window: a line never spans two */
        else if (s && n) str_append(s, (const char *)p, n);
        got += (long)n; r->beg += n + (nl ? 1 : 0);
        if (nl) return got;
    }
}
static th_reader_t *rd_open(const char *fn)
{
    th_reader_t *r = (th_reader_t *)calloc(1, sizeof(*r));
    r->scur = -1; r->gcur = -1;
    int fd = strcmp(fn, "-") ? open(fn, O_RDONLY) : -1;
    struct stat st;
    if (strcmp(fn, "-") && fd < 0) { free(r); return NULL; }
    if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && !getenv("TH_READER_STREAM")) {
        void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            const unsigned char *p = (const unsigned char *)m;
            uint32_t cl, is; const unsigned char *cd;
            if (st.st_size >= 2 && p[0] == 0x1f && p[1] == 0x8b) {
                if (bgzf_member(p, (size_t)st.st_size, &cl, &is, &cd)) {
                    r->mode = RD_BGZF; r->map = p; r->map_len = (size_t)st.st_size;
                    r->blk = (bgzf_blk_t *)malloc(sizeof(bgzf_blk_t) * BGZF_GROUP);
                    pthread_mutex_init(&r->zmu, NULL); pthread_cond_init(&r->zcv, NULL);
                    const int nc = th_effective_cpus(); r->nz = nc >= 32 ? 12 : nc >= 8 ? 6 : 2;       /* (the CPUs the process may use, not the host's) */
                    { const char *e = getenv("TH_READER_THREADS"); if (e && atoi(e) >= 1 && atoi(e) <= 16) r->nz = atoi(e); }
                    { int made = 0; for (int t = 0; t < r->nz; ++t) if (pthread_create(&r->zthr[made], NULL, bgzf_worker, r) == 0) ++made; r->nz = made; }
                    if (r->nz == 0) { pthread_mutex_destroy(&r->zmu); pthread_cond_destroy(&r->zcv); free(r->blk); munmap(m, (size_t)st.st_size); close(fd); free(r); return NULL; }
                    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                    close(fd);
                    return r;
                }
                munmap(m, (size_t)st.st_size);           /* plain gzip: stream it */
            } else {
                r->mode = RD_MEM; r->map = p; r->map_len = (size_t)st.st_size; r->buf = p; r->beg = 0; r->end = (size_t)st.st_size;
                madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                r->lx = lx_open(p, (size_t)st.st_size);
                close(fd);
                return r;
            }
        }
    }
    r->mode = RD_STREAM;
    r->fp = fd >= 0 ? gzdopen(fd, "r") : gzdopen(0, "r");
    if (!r->fp) { if (fd >= 0) close(fd); free(r); return NULL; }
    gzbuffer(r->fp, 1 << 20);
    r->sbuf[0] = (unsigned char *)malloc(RD_CHUNK); r->sbuf[1] = (unsigned char *)malloc(RD_CHUNK);
    pthread_mutex_init(&r->smu, NULL); pthread_cond_init(&r->scv, NULL);
    if (pthread_create(&r->sthr, NULL, stream_producer, r) != 0) { gzclose(r->fp); free(r->sbuf[0]); free(r->sbuf[1]); free(r); return NULL; }
    r->sthr_on = 1;
    return r;
}
static void rd_close(th_reader_t *r)
{
    if (!r) return;
    if (r->mode == RD_STREAM) {
        if (r->sthr_on) {          /* the producer finishes the read it is in, sees the flag and stops */
            pthread_mutex_lock(&r->smu); r->squit = 1; pthread_cond_broadcast(&r->scv); pthread_mutex_unlock(&r->smu);
            pthread_join(r->sthr, NULL);
        }
        gzclose(r->fp); free(r->sbuf[0]); free(r->sbuf[1]);
    } else {
        if (r->mode == RD_BGZF) {
            bgzf_wait_group(r);
            pthread_mutex_lock(&r->zmu); r->zquit = 1; pthread_cond_broadcast(&r->zcv); pthread_mutex_unlock(&r->zmu);
            for (int t = 0; t < r->nz; ++t) pthread_join(r->zthr[t], NULL);
            free(r->gbuf[0]); free(r->gbuf[1]); free(r->blk);
        }
        lx_close(r->lx);
        munmap((void *)r->map, r->map_len);
    }
    free(r);
}
/* Appends one record to the arena `a` as name\0[sequence\0] and returns its offsets (the arena may move while it grows, so
 * callers turn offsets into pointers when the batch is complete).  A sequence that sits on one line of a mapped file is not
 * copied: *zc_seq points into the mapping and *seq_off is (size_t)-1.  1 = record read, 0 = end of file.  One arena per batch
 * slot, reused from batch to batch: no per-read malloc/free, and no page-fault storm next to the device allocations. */
static int rd_next_into(th_reader_t *r, th_str_t *a, size_t *name_off, size_t *seq_off, const char **zc_seq, int *len)
{
    int c;
    if (zc_seq) *zc_seq = NULL;
    if (r->last == 0) {
        while ((c = rd_getc(r)) != -1 && c != '>' && c != '@') {}
        if (c == -1) return 0;
        r->last = c;
    }
    *name_off = a->l;
    while ((c = rd_getc(r)) != -1 && !isspace(c)) str_putc(a, c);
    str_putc(a, 0);
    if (c != '\n' && c != -1) while ((c = rd_getc(r)) != -1 && c != '\n') {}
    *seq_off = a->l;
    const char *zc = NULL; size_t zn = 0, sl = 0; int n_lines = 0;
    const int want_zc = zc_seq != NULL && r->mode == RD_MEM;
    while ((c = rd_getc(r)) != -1 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        --r->beg;                                         /* the line starts with c: take it whole */
        if (n_lines == 1 && zc) { str_append(a, zc, zn); zc = NULL; }      /* a second line: the sequence has to be assembled */
        const char *lz = NULL; size_t ln = 0;
        const size_t before = a->l;
        rd_line(r, a, (want_zc && n_lines == 0) ? &lz : NULL, &ln);
        if (lz) { zc = lz; zn = ln; sl = ln; if (sl > 1 && zc[zn - 1] == '\r') { --zn; --sl; } }
        else { sl += a->l - before; if (sl > 1 && a->s[a->l - 1] == '\r') { a->s[--a->l] = 0; --sl; } }
        ++n_lines;
    }
    if (zc) { *zc_seq = zc; *seq_off = (size_t)-1; }
    else str_putc(a, 0);
    r->last = (c == '>' || c == '@') ? c : 0;
    if (c == '+') {
        while ((c = rd_getc(r)) != -1 && c != '\n') {}
        size_t ql = 0; long n;
        while (ql < sl && (n = rd_line(r, NULL, NULL, NULL)) >= 0) ql += (size_t)n;   /* quality lines until as long as the sequence */
        r->last = 0;
    }
    *len = (int)sl;
    return 1;
}
/* 1 = record read (name/seq malloc'd), 0 = end of file */
static int rd_next(th_reader_t *r, char **name, char **seq, int *len)
{
    th_str_t a = { 0, 0, 0 }; size_t no, so;
    if (!rd_next_into(r, &a, &no, &so, NULL, len)) { free(a.s); return 0; }
    *name = strdup(a.s + no); *seq = (char *)malloc((size_t)*len + 1); memcpy(*seq, a.s + so, (size_t)*len); (*seq)[*len] = 0;
    free(a.s);
    return 1;
}

static char *rc_seq(const char *s, int len)   /* get_rc_seq, src/seq.c:88-95 */
{
    char *rc = (char *)malloc((size_t)len + 1);
    for (int i = 0; i < len; ++i) {
        char c = s[i], o = 'N';
        switch (c) {
        case 'A': case 'a': case 0: o = 'T'; break;
        case 'C': case 'c': case 1: o = 'G'; break;
        case 'G': case 'g': case 2: o = 'C'; break;
        case 'T': case 't': case 3: o = 'A'; break;
        default: o = 'N';
        }
        rc[len - 1 - i] = o;
    }
    rc[len] = 0;
    return rc;
}

static int load_first_seq(const char *fn, char **seq)
{
    th_reader_t *r = rd_open(fn);
    if (!r) { char m[300]; snprintf(m, sizeof(m), "fail to open file '%s'", fn); host_fail(m); return -1; }
    char *nm = NULL, *sq = NULL; int len = 0;
    if (rd_next(r, &nm, &sq, &len) && len > 0) { *seq = sq; free(nm); }
    else { fprintf(stderr, "[get_seq_from_fx] No sequence found.\n"); free(nm); free(sq); len = 0; }
    rd_close(r);
    return len;
}

int th_load_adapters(th_para_t *p)
{
    if (!p->five_fn || !p->three_fn) return 0;
    if ((p->five_len = load_first_seq(p->five_fn, &p->five_seq)) < 0) return -1;
    p->five_rc_seq = rc_seq(p->five_seq ? p->five_seq : "", p->five_len);
    if ((p->three_len = load_first_seq(p->three_fn, &p->three_seq)) < 0) return -1;
    p->three_rc_seq = rc_seq(p->three_seq ? p->three_seq : "", p->three_len);
    return 0;
}

/* ---------------------------------------------------------------- tandem_seq_t */
th_tandem_seq_t *th_alloc_tandem_seq(int n)
{
    th_tandem_seq_t *t = (th_tandem_seq_t *)calloc((size_t)n, sizeof(th_tandem_seq_t));
    for (int i = 0; i < n; ++i) {
        t[i].cons_seq = (th_seq_t *)calloc(1, sizeof(th_seq_t));
        t[i].cons_m = 1;
        t[i].cons_start = (int *)malloc(sizeof(int)); t[i].cons_end = (int *)malloc(sizeof(int)); t[i].cons_len = (int *)malloc(sizeof(int));
        t[i].copy_num = (double *)malloc(sizeof(double)); t[i].ave_match = (double *)malloc(sizeof(double));
        t[i].full_length = (int8_t *)malloc(1); t[i].cons_score = (int *)malloc(sizeof(int));
        t[i].pos_n = (int *)calloc(1, sizeof(int)); t[i].pos_m = (int *)calloc(1, sizeof(int)); t[i].sub_pos = (int **)calloc(1, sizeof(int *));
    }
    return t;
}
void th_free_tandem_seq(th_tandem_seq_t *t, int n)
{
    if (!t) return;
    for (int i = 0; i < n; ++i) {
        free(t[i].cons_seq->name.s); free(t[i].cons_seq->comment.s); free(t[i].cons_seq->seq.s); free(t[i].cons_seq->qual.s); free(t[i].cons_seq);
        free(t[i].cons_start); free(t[i].cons_end); free(t[i].cons_len);
        free(t[i].copy_num); free(t[i].ave_match); free(t[i].full_length); free(t[i].cons_score); free(t[i].pos_n); free(t[i].pos_m);
        for (int j = 0; j < t[i].cons_m; ++j) free(t[i].sub_pos[j]);
        free(t[i].sub_pos);
    }
    free(t);
}
static void tseq_grow(th_tandem_seq_t *t)
{
    if (t->cons_n < t->cons_m) return;
    int old = t->cons_m; t->cons_m <<= 1;
    t->cons_start = (int *)realloc(t->cons_start, t->cons_m * sizeof(int)); t->cons_end = (int *)realloc(t->cons_end, t->cons_m * sizeof(int));
    t->copy_num = (double *)realloc(t->copy_num, t->cons_m * sizeof(double)); t->ave_match = (double *)realloc(t->ave_match, t->cons_m * sizeof(double));
    t->full_length = (int8_t *)realloc(t->full_length, t->cons_m); t->cons_len = (int *)realloc(t->cons_len, t->cons_m * sizeof(int));
    t->cons_score = (int *)realloc(t->cons_score, t->cons_m * sizeof(int)); t->sub_pos = (int **)realloc(t->sub_pos, t->cons_m * sizeof(int *));
    t->pos_n = (int *)realloc(t->pos_n, t->cons_m * sizeof(int)); t->pos_m = (int *)realloc(t->pos_m, t->cons_m * sizeof(int));
    for (int i = old; i < t->cons_m; ++i) { t->pos_m[i] = 0; t->pos_n[i] = 0; t->sub_pos[i] = NULL; }
}
static void tseq_set_pos(th_tandem_seq_t *t, const int *pos, int pos_n)
{
    int c = t->cons_n;
    t->pos_n[c] = pos_n;
    if (pos_n > t->pos_m[c]) { t->pos_m[c] = pos_n; t->sub_pos[c] = (int *)realloc(t->sub_pos[c], pos_n * sizeof(int)); }
    for (int i = 0; i < pos_n; ++i) t->sub_pos[c][i] = pos[i];
}
/* write_tandem_cons_seq, src/gen_cons.c:10-62.  (Unlike the reference, the quality buffer is rewound
 * together with the sequence buffer: see DESIGN.md "reference bugs not mirrored".) */
static void write_cons(th_tandem_seq_t *t, const char *seq, const uint8_t *qual, int cons_len, int start, int end, double copy_num,
                       double ave_match, const th_para_t *p, int full_length, const int *pos, int pos_n)
{
    if (cons_len < p->min_len || cons_len > p->max_p) return;
    if (p->only_longest && t->cons_n == 1) {
        if (end - start > t->cons_end[0] - t->cons_start[0]) { t->cons_n = 0; t->cons_seq->seq.l = 0; t->cons_seq->qual.l = 0; }
        else return;
    }
    th_kstring_t *cs = &t->cons_seq->seq, *cq = &t->cons_seq->qual;
    if (cs->l + cons_len >= cs->m) { cs->m = cs->l + cons_len + 1; cs->s = (char *)realloc(cs->s, cs->m); }
    memcpy(cs->s + cs->l, seq, cons_len); cs->l += cons_len;
    if (qual) {
        if (cq->l + cons_len >= cq->m) { cq->m = cq->l + cons_len + 1; cq->s = (char *)realloc(cq->s, cq->m); }
        memcpy(cq->s + cq->l, qual, cons_len); cq->l += cons_len;
    }
    tseq_grow(t);
    int c = t->cons_n;
    t->cons_start[c] = start; t->cons_end[c] = end; t->copy_num[c] = copy_num; t->full_length[c] = (int8_t)full_length;
    t->cons_len[c] = cons_len; t->ave_match[c] = ave_match;
    tseq_set_pos(t, pos, pos_n);
    ++t->cons_n;
}
static void write_unit(th_tandem_seq_t *t, const int *pos, int pos_n) { tseq_grow(t); tseq_set_pos(t, pos, pos_n); ++t->cons_n; }   /* src/gen_cons.c:64-83 */

/* ---------------------------------------------------------------- edlib HW/LOC: Myers' block bit-vector, own implementation */
static int eq_char(char a, char b)   /* byte equality + a/A c/C g/G t/T n/N (src/edlib_align.c:21-27) */
{
    if (a == b) return 1;
    if ((a ^ b) != 0x20) return 0;
    char x = a | 0x20;
    return x == 'a' || x == 'c' || x == 'g' || x == 't' || x == 'n';
}
/* One pass of Myers' algorithm over `t` (read with stride ts), query bits precomputed per text byte.
 * hw: 1 = infix (free leading text), 0 = prefix (SHW).  Calls back per column with the score. */
typedef struct { int nb, m; uint64_t *peq; /* [256][nb] */ } myers_t;
static myers_t *myers_build(const char *q, int qs, int m)
{
    myers_t *my = (myers_t *)calloc(1, sizeof(*my));
    my->m = m; my->nb = (m + 63) / 64;
    my->peq = (uint64_t *)calloc((size_t)256 * my->nb, sizeof(uint64_t));
    for (int c = 0; c < 256; ++c)
        for (int i = 0; i < m; ++i)
            if (eq_char(q[(long)i * qs], (char)c)) my->peq[(size_t)c * my->nb + i / 64] |= 1ull << (i & 63);
    return my;
}
static void myers_free(myers_t *my) { free(my->peq); free(my); }
/* returns scores per column in sc[0..n) */
static void myers_run(const myers_t *my, const char *t, int ts, int n, int hw, int *sc)
{
    int nb = my->nb, m = my->m, lastbit = (m - 1) & 63, score = m;
    uint64_t *Pv = (uint64_t *)malloc(sizeof(uint64_t) * nb * 2), *Mv = Pv + nb;
    for (int b = 0; b < nb; ++b) { Pv[b] = ~0ull; Mv[b] = 0; }
    for (int c = 0; c < n; ++c) {
        const uint64_t *peq = my->peq + (size_t)(unsigned char)t[(long)c * ts] * nb;
        int hin = hw ? 0 : 1;
        for (int b = 0; b < nb; ++b) {
            uint64_t Eq = peq[b], pv = Pv[b], mv = Mv[b];
            uint64_t Xv = Eq | mv;
            if (hin < 0) Eq |= 1;
            uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint64_t Ph = mv | ~(Xh | pv), Mh = pv & Xh;
            int hout;
            if (b == nb - 1) { hout = (int)((Ph >> lastbit) & 1) - (int)((Mh >> lastbit) & 1); score += hout; }
            else hout = (int)(Ph >> 63) - (int)(Mh >> 63);
            Ph <<= 1; Mh <<= 1;
            if (hin < 0) Mh |= 1; else if (hin > 0) Ph |= 1;
            Pv[b] = Mh | ~(Xv | Ph); Mv[b] = Ph & Xv;
            hin = hout;
        }
        sc[c] = score;
    }
    free(Pv);
}
/* edlib_align_HW: distance, first best end, and edlib's start for it; -1 if the distance exceeds k */
static int edlib_hw(const char *q, int qlen, const char *t, int tlen, int *start, int *end, int k)
{
    if (qlen <= 0 || tlen <= 0) return -1;
    int *sc = (int *)malloc(sizeof(int) * (size_t)tlen);
    myers_t *my = myers_build(q, 1, qlen);
    myers_run(my, t, 1, tlen, 1, sc);
    myers_free(my);
    int best = sc[0], be = 0;
    for (int c = 1; c < tlen; ++c) if (sc[c] < best) { best = sc[c]; be = c; }
    if (k >= 0 && best > (k < qlen ? k : qlen)) { free(sc); return -1; }
    if (best >= qlen && (64 - qlen % 64) % 64 > 0) { *start = 0; *end = -1; free(sc); return qlen; }   /* edlib's "position -1" */
    /* start: reversed query against the reversed target prefix, prefix mode, last best column */
    my = myers_build(q + qlen - 1, -1, qlen);
    myers_run(my, t + be, -1, be + 1, 0, sc);
    myers_free(my);
    int last_p = -1;
    for (int c = 0; c <= be; ++c) if (sc[c] == best) last_p = c;
    free(sc);
    *end = be; *start = be - last_p;
    return best;
}

/* ---------------------------------------------------------------- batch seam */
static th_handle_t *th_init_sub(const th_para_t *p, int device, int n_sub_max)
{
    g_err[0] = 0;
    if (p->w < 1 || p->w > 256) { host_fail("-w must be in 1..256 (the reference's minimizer window, src/tandem_hit.c:100, has 256 slots)"); return NULL; }
    if (p->k < 2 || p->k > 16) { host_fail("k-mer length must be in [2,16]"); return NULL; }
    th_gpu_para_t gp; memset(&gp, 0, sizeof(gp));
    gp.k = (int)p->k; gp.w = (int)p->w; gp.hpc = p->hpc ? 1 : 0; gp.min_p = (uint32_t)p->min_p; gp.max_p = (uint32_t)p->max_p; gp.one_minus_div = 1 - p->max_div; gp.min_copy = p->min_copy;
    gp.match = p->match; gp.mismatch = p->mismatch; gp.go1 = p->gap_open1; gp.ge1 = p->gap_ext1; gp.go2 = p->gap_open2; gp.ge2 = p->gap_ext2;
    gp.only_unit = p->only_unit;
    gp.single_copy = (p->single_copy == 1 && p->only_full_length) ? 1 : 0;
    gp.want_cov = (p->min_cov > 0 || p->min_frac > 0.0 || p->out_fmt == TH_FASTQ_FMT || p->out_fmt == TH_TAB_QUAL_FMT) ? 1 : 0;
    if (p->five_seq != NULL && p->three_seq != NULL && p->five_rc_seq != NULL && p->three_rc_seq != NULL) {   /* src/gen_cons.c:226,238-263 */
        gp.ad_seq[0] = p->five_seq; gp.ad_seq[1] = p->three_rc_seq; gp.ad_seq[2] = p->five_rc_seq; gp.ad_seq[3] = p->three_seq;
        gp.ad_len[0] = gp.ad_len[2] = p->five_len; gp.ad_len[1] = gp.ad_len[3] = p->three_len;
        gp.ad_k[0] = gp.ad_k[2] = (int)(p->five_len * (1 - p->ada_match_rat)); gp.ad_k[1] = gp.ad_k[3] = (int)(p->three_len * (1 - p->ada_match_rat));
    }
    th_handle_t *h = (th_handle_t *)calloc(1, sizeof(*h));
    h->para = *p; h->n_sub_max = n_sub_max < 1 ? 1 : n_sub_max > TH_MAX_SUB ? TH_MAX_SUB : n_sub_max; h->up[0].n_sub = h->up[1].n_sub = h->res[0].n_sub = h->res[1].n_sub = 1;
    h->sub_min = 16384;
    { const char *e = getenv("TH_SUBBATCH_MIN"); if (e && atoi(e) > 0) h->sub_min = atoi(e); }
    for (int s = 0; s < h->n_sub_max; ++s) {
        h->gpu[s] = th_gpu_create(&gp, device);
        if (!h->gpu[s]) { for (int x = 0; x < s; ++x) th_gpu_destroy(h->gpu[x]); free(h); return NULL; }
    }
    __atomic_fetch_add(&g_live_handles, 1, __ATOMIC_RELAXED);
    return h;
}
th_handle_t *th_init(const th_para_t *p, int device)
{
    int ns = 1;          /* one pipeline, one stream (TH_SUBBATCH=2: a large batch as two concurrent halves, round 2's default) */
    const char *e = getenv("TH_SUBBATCH");
    if (e && atoi(e) >= 1) ns = atoi(e);
    return th_init_sub(p, device, ns);
}
void th_destroy(th_handle_t *h) { if (!h) return; for (int s = 0; s < h->n_sub_max; ++s) th_gpu_destroy(h->gpu[s]); __atomic_fetch_sub(&g_live_handles, 1, __ATOMIC_RELAXED); free(h); }

static int batch_upload_s(th_handle_t *h, int slot, const char *const *seqs, const int *lens, int n, int wait)
{
    th_batch_t *b = &h->up[slot & 1];
    b->n = n; b->seqs = seqs; b->lens = lens;
    b->n_sub = (h->n_sub_max >= 2 && n >= h->sub_min) ? h->n_sub_max : 1;
    b->sub_off[0] = 0; b->sub_off[b->n_sub] = n;
    if (b->n_sub > 1) {                        /* parts balanced by bases (cost ~ bases) */
        long long tot = 0, acc = 0; int cut = 0;
        for (int i = 0; i < n; ++i) tot += lens[i];
        for (int s = 1; s < b->n_sub; ++s) {
            while (cut < n - (b->n_sub - s) && (acc + lens[cut]) * b->n_sub <= tot * s) acc += lens[cut++];
            if (cut < s) { acc += lens[cut]; ++cut; }          /* never an empty part */
            b->sub_off[s] = cut;
        }
    }
    for (int s = 0; s < b->n_sub; ++s)
        if (th_gpu_upload(h->gpu[s], slot, seqs + b->sub_off[s], lens + b->sub_off[s], b->sub_off[s + 1] - b->sub_off[s], wait)) return -1;
    return 0;
}
int th_batch_upload(th_handle_t *h, const char *const *seqs, const int *lens, int n) { return batch_upload_s(h, 0, seqs, lens, n, 1); }
int th_batch_reserve(th_handle_t *h, int max_reads, long long max_bases)
{
    const int ns = (h->n_sub_max >= 2 && max_reads >= h->sub_min) ? h->n_sub_max : 1;
    for (int s = 0; s < ns; ++s) if (th_gpu_reserve(h->gpu[s], max_reads / ns + 64, max_bases / ns + max_bases / 16)) return -1;
    return 0;
}
/* (Tried: starting the second half only when the first has finished stages 1-3, so that the halves sit in different stages.  Worse:
 * the late half's first kernels queue behind the POA grid -- config 3 88.1 k against 92.0 k reads/s, config 2 171 k against 185 k.) */
typedef struct { th_gpu_t *g; int in_slot, out_slot; } sub_run_arg_t;
static void *sub_run(void *arg) { sub_run_arg_t *a = (sub_run_arg_t *)arg; return (void *)(intptr_t)th_gpu_run(a->g, a->in_slot, a->out_slot); }
static int batch_run_s(th_handle_t *h, int in_slot, int out_slot)
{
    h->res[out_slot & 1] = h->up[in_slot & 1];
    const int n_sub = h->res[out_slot & 1].n_sub;
    if (n_sub == 1) return th_gpu_run(h->gpu[0], in_slot, out_slot);
    pthread_t tid[TH_MAX_SUB]; sub_run_arg_t arg[TH_MAX_SUB]; int bad = 0;
    for (int s = 1; s < n_sub; ++s) {
        arg[s].g = h->gpu[s]; arg[s].in_slot = in_slot; arg[s].out_slot = out_slot;
        if (pthread_create(&tid[s], NULL, sub_run, &arg[s]) != 0) { for (int x = 1; x < s; ++x) pthread_join(tid[x], NULL); return host_fail("pthread_create failed"); }
    }
    const int r0 = th_gpu_run(h->gpu[0], in_slot, out_slot);
    for (int s = 1; s < n_sub; ++s) { void *r = NULL; pthread_join(tid[s], &r); if (r != NULL) bad = 1; }
    if (bad && !g_err[0]) snprintf(g_err, sizeof(g_err), "a part of the batch failed on the device (see stderr)");
    return (r0 || bad) ? -1 : 0;
}
int th_batch_run(th_handle_t *h) { return batch_run_s(h, 0, 0); }
/* device milliseconds of the last run per kernel, summed over the concurrent parts; th_batch_launches() says how many there were */
int th_batch_timing(th_handle_t *h, double *ms_kernel, int n)
{
    for (int i = 0; i < n && i < 8; ++i) ms_kernel[i] = 0;
    for (int s = 0; s < h->res[0].n_sub; ++s) {
        th_gpu_out_t o; th_gpu_get(h->gpu[s], 0, &o);
        for (int i = 0; i < n && i < 8; ++i) ms_kernel[i] += o.ms_kernel[i];
    }
    return 6;
}
int th_batch_launches(th_handle_t *h) { return h->res[0].n_sub; }
int th_batch_syncs(th_handle_t *h) { return th_gpu_syncs(h->gpu[0]); }
int th_batch_stats(th_handle_t *h, long long *out, int n)
{
    for (int i = 0; i < n && i < 8; ++i) out[i] = 0;
    for (int s = 0; s < TH_MAX_SUB; ++s) if (h->gpu[s]) { long long v[8] = { 0 }; th_gpu_stats(h->gpu[s], v, 8); for (int i = 0; i < n && i < 8; ++i) out[i] += v[i]; }
    return 5;
}
long long th_debug_stage(th_handle_t *h, int what, void *dst, long long cap) { return h->res[0].n_sub == 1 ? th_gpu_debug(h->gpu[0], what, dst, cap) : -1; }

typedef struct { int ed, start, end; } ed_res_t;
static int collect_ed(const th_para_t *p, const char *q, int qlen, const char *seq, int seq_len, ed_res_t *r)   /* src/gen_cons.c:89-110 */
{
    int n = 0, ed, s = 0, e = 0, k = (int)(qlen * (1 - p->ada_match_rat));
    ed = edlib_hw(q, qlen, seq, seq_len, &s, &e, k);
    if (ed != -1) {
        r[0].ed = ed; r[0].start = s; r[0].end = e; n = 1;
        if (r[0].start >= qlen) { ed = edlib_hw(q, qlen, seq, r[0].start, &s, &e, k); if (ed != -1) { r[n].ed = ed; r[n].start = s; r[n].end = e; ++n; } }
        if (r[0].end <= seq_len - qlen) {
            ed = edlib_hw(q, qlen, seq + r[0].end, seq_len - r[0].end, &s, &e, k);
            if (ed != -1) { r[n].ed = ed; r[n].start = r[0].end + s; r[n].end = r[0].end + e; ++n; }
        }
    }
    return n;
}
static int full_len_pick(const th_para_t *p, int ln, const ed_res_t *l, int rn, const ed_res_t *r, int *ts, int *te)   /* :112-126 */
{
    int tot = INT_MAX;
    for (int i = 0; i < ln; ++i) for (int j = 0; j < rn; ++j)
        if (r[j].start - l[i].end - 1 >= p->min_len && tot > l[i].ed + r[j].ed) { tot = l[i].ed + r[j].ed; *ts = l[i].end + 1; *te = r[j].start - 1; }
    return tot;
}
/* dev: the searches as the device ran them (th_k_adapter_reads: 10 ints per query -- 5', rc(3'), rc(5'), 3' -- hit count and up to three
 * (ed, start, end)), or NULL: then the host searches (adapters of more than 256 symbols) */
static int dev_ed(const int *dev, int query, ed_res_t *r)
{
    const int *o = dev + 10 * query; const int n = o[0];
    for (int i = 0; i < n; ++i) { r[i].ed = o[1 + 3 * i]; r[i].start = o[2 + 3 * i]; r[i].end = o[3 + 3 * i]; }
    return n;
}
static void single_copy(int seq_len, const char *seq, th_tandem_seq_t *t, const th_para_t *p, const int *dev)   /* :128-171 */
{
    int cons_len = 0, full_length = 0, ts = -1, te = -1, tot, pos[2] = { 0, 0 };
    ed_res_t r5[3], r3[3];
    int n5 = dev ? dev_ed(dev, 0, r5) : collect_ed(p, p->five_seq, p->five_len, seq, seq_len, r5);
    int n3 = dev ? dev_ed(dev, 1, r3) : collect_ed(p, p->three_rc_seq, p->three_len, seq, seq_len, r3);
    tot = full_len_pick(p, n5, r5, n3, r3, &ts, &te);
    if (tot != INT_MAX) { pos[0] = ts; pos[1] = te; cons_len = te - ts + 1; full_length = 1; }
    if (tot > 0) {
        n5 = dev ? dev_ed(dev, 2, r5) : collect_ed(p, p->five_rc_seq, p->five_len, seq, seq_len, r5);
        n3 = dev ? dev_ed(dev, 3, r3) : collect_ed(p, p->three_seq, p->three_len, seq, seq_len, r3);
        if (full_len_pick(p, n3, r3, n5, r5, &ts, &te) < tot) { pos[0] = ts; pos[1] = te; cons_len = te - ts + 1; full_length = 2; }
    }
    if (full_length > 0) {
        if (p->only_unit) write_unit(t, pos, 2);
        else {
            uint8_t *q = NULL;
            if (p->out_fmt == TH_FASTQ_FMT || p->out_fmt == TH_TAB_QUAL_FMT) { q = (uint8_t *)malloc((size_t)cons_len + 1); memset(q, 33, cons_len); }
            write_cons(t, seq + pos[0], q, cons_len, pos[0], pos[1], 1.0, 100.0, p, full_length, pos, 2);
            free(q);
        }
    }
}

static int code_of(unsigned char c)
{
    switch (c) { case 0: case 'A': case 'a': return 0; case 1: case 'C': case 'c': return 1; case 2: case 'G': case 'g': return 2;
                 case 3: case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

/* per-thread scratch of the record arithmetic: four byte buffers that only ever grow (a malloc / free pair per job and buffer was a
 * measurable share of the collecting threads' time at a million jobs per 8 s) */
typedef struct { char *b[4]; size_t m[4]; } th_scratch_t;
static void *scr(th_scratch_t *sc, int which, size_t n)
{
    if (n > sc->m[which]) { sc->m[which] = n + (n >> 2) + 256; free(sc->b[which]); sc->b[which] = (char *)malloc(sc->m[which]); }
    return sc->b[which];
}
static void scr_free(th_scratch_t *sc) { for (int i = 0; i < 4; ++i) { free(sc->b[i]); sc->b[i] = NULL; sc->m[i] = 0; } }

/* host half of seqs_msa()/abpoa_gen_cons() for one run, from the integer evidence the device produced */
static void finish_job(const th_handle_t *h, const th_batch_t *B, const th_gpu_out_t *o, const th_job_t *j, th_tandem_seq_t *t, int rb, th_scratch_t *sc)
{
    const th_para_t *p = &h->para;
    const int *ext = o->ext + 4 * (j - o->jobs);            /* max_q_l, max_t_l, max_q_r, max_t_r */
    const int *pos = o->par + j->pos_off; const int pos_n = j->pos_n;
    const char *seq = B->seqs[rb + j->read]; const int seq_len = B->lens[rb + j->read];      /* rb: first read of the half this job belongs to */
    if (p->only_unit) { write_unit(t, pos, pos_n); return; }
    int cons_len = j->cons_len, n_seqs = j->n_seqs;
    if (cons_len <= 0) return;
    const uint8_t *cb = o->cons + j->cons_off; const int *cov = o->cov ? o->cov + j->cons_off : NULL;
    const int want_q = p->out_fmt == TH_FASTQ_FMT || p->out_fmt == TH_TAB_QUAL_FMT;
    int min_cov = 0;
    if (p->min_frac > 0.0) min_cov = (int)(n_seqs * p->min_frac); else if (p->min_cov > 0) min_cov = p->min_cov;
    uint8_t *qual = want_q ? (uint8_t *)scr(sc, 0, (size_t)cons_len * 2 + 2) : NULL;
    if (n_seqs <= 2) {                                             /* src/abpoa_cons.c:60-83 */
        if (min_cov > 0) {
            /* the first two units that pass the filter at :43 */
            int u[2], nu = 0;
            for (int i = 0; i < pos_n - 1 && nu < 2; ++i) {
                int s = pos[i], e = pos[i + 1];
                if (s < 0 || e < 0 || s >= seq_len - 1 || e + 1 > seq_len) continue;
                u[nu++] = i;
            }
            int mc = 2, l0 = pos[u[0] + 1] - pos[u[0]], l1 = pos[u[1] + 1] - pos[u[1]];
            if (l0 != l1) mc = 1;
            else for (int i = 0; i < l0; ++i) if (code_of((unsigned char)seq[pos[u[0]] + 1 + i]) != code_of((unsigned char)seq[pos[u[1]] + 1 + i])) { mc = 1; break; }
            if (mc < min_cov) return;
        }
        if (qual) memset(qual, 33, cons_len);
    } else {                                                       /* :88-108 */
        if (min_cov > 0) for (int i = 0; i < cons_len; ++i) if (cov[i] < min_cov) return;
        if (qual) for (int i = 0; i < cons_len; ++i) {
            double x = 13.8 * (1.25 * cov[i] / n_seqs - 0.25);
            double pe = 1 - 1.0 / (1.0 + pow(2.718281828459045, -1 * x));
            qual[i] = (uint8_t)(33 + (int)(-10 * log10(pe) + 0.499));
        }
    }
    double ave = 0;                                                /* src/gen_cons.c:208-214 */
    for (int k = 0; k < pos_n - 1; ++k) { int len = pos[k + 1] - pos[k]; ave += (o->iden[j->aln_off + k] * 100 / (len + 0.0)); }
    char *cons_seq = (char *)scr(sc, 1, (size_t)cons_len * 2 + 2);
    for (int s = 0; s < cons_len; ++s) cons_seq[s] = cb[s] <= 4 ? "ACGTN"[cb[s]] : 'N';
    cons_seq[cons_len] = 0;
    double copy_num = n_seqs;                                      /* :217-223 */
    int cons_start = pos[0] - ext[1];
    copy_num += (ext[0] + 1.0) / cons_len;
    int cons_end = pos[pos_n - 1] + ext[3] + 1;
    copy_num += (ext[2] + 1.0) / cons_len;
    int full_length = 0, out_len = cons_len;
    if (p->five_seq != NULL && p->three_seq != NULL && cons_len > p->five_len + p->three_len) {   /* :226-291 */
        char *cons2 = (char *)scr(sc, 2, ((size_t)cons_len << 1) + 1); uint8_t *qual2 = NULL;
        memcpy(cons2, cons_seq, cons_len); memcpy(cons2 + cons_len, cons_seq, cons_len); cons2[cons_len << 1] = 0;
        if (qual) { qual2 = (uint8_t *)scr(sc, 3, (size_t)cons_len << 1); memcpy(qual2, qual, cons_len); memcpy(qual2 + cons_len, qual, cons_len); }
        int tar_start = -1, tar_end = -1, tot_ed = INT_MAX, e5, e3, s5 = -1, n5 = -1, s3 = -1, n3 = -1;
        int k5 = (int)(p->five_len * (1 - p->ada_match_rat)), k3 = (int)(p->three_len * (1 - p->ada_match_rat));
        /* the four searches ran on the device (th_k_adapter) when o->adapt is set: (ed, start, end) of 5', rc(3'), rc(5'), 3' */
        const int *ad = o->adapt ? o->adapt + 12 * (int)(j - o->jobs) : NULL;
#define AD_HW(sr, q, ql, kk, st, en) (ad ? (*(st) = ad[3 * (sr) + 1], *(en) = ad[3 * (sr) + 2], ad[3 * (sr)]) : edlib_hw((q), (ql), cons2, cons_len << 1, (st), (en), (kk)))
        e5 = AD_HW(0, p->five_seq, p->five_len, k5, &s5, &n5);
        if (e5 == -1) goto REV;
        e3 = AD_HW(1, p->three_rc_seq, p->three_len, k3, &s3, &n3);
        if (e3 == -1) goto REV;
        if (s3 <= n5) {
            if (n3 + cons_len < cons_len << 1 && s3 + cons_len > n5) { tar_start = n5 + 1; tar_end = s3 + cons_len - 1; full_length = 1; tot_ed = e5 + e3; }
        } else { tar_start = n5 + 1; tar_end = s3 - 1; tot_ed = e5 + e3; full_length = 1; }
        if (tot_ed == 0) goto WRITE;
REV:
        e5 = AD_HW(2, p->five_rc_seq, p->five_len, k5, &s5, &n5);
        if (e5 == -1) goto WRITE;
        e3 = AD_HW(3, p->three_seq, p->three_len, k3, &s3, &n3);
        if (e3 == -1) goto WRITE;
#undef AD_HW
        if (e5 + e3 < tot_ed) {
            if (s5 <= n3) {
                if (n5 + cons_len < cons_len << 1 && s5 + cons_len > n3) { tar_start = n3 + 1; tar_end = s5 + cons_len - 1; full_length = 2; }
            } else { tar_start = n3 + 1; tar_end = s5 - 1; full_length = 2; }
        }
WRITE:
        if (tar_start > 0 && tar_end > tar_start) {
            out_len = tar_end - tar_start + 1;
            memcpy(cons_seq, cons2 + tar_start, out_len); cons_seq[out_len] = 0;
            if (qual) for (int k = tar_start; k <= tar_end; ++k) qual[k - tar_start] = qual2[k];
        }
    }
    if (!p->only_full_length || full_length > 0)
        write_cons(t, cons_seq, qual, out_len, cons_start, cons_end, copy_num, ave / (pos_n - 1), p, full_length, pos, pos_n);
}

/* ---------------------------------------------------------------- output, src/main.c:214-271 */
/* The records of one read as text, appended to a growing buffer.  The file driver formats inside the threads that do the record
 * arithmetic (a range of reads each, in read order) and its writer only hands the finished buffers to fwrite; th_output() formats
 * and writes in one go.  The reference prints with fprintf, a "%c" per base (src/main.c:225,232,255,260): one writer thread at
 * ~1 microsecond per record is exactly the 10^6 records/s eight GPUs deliver, so nothing per record may sit on the writer. */
static inline void sb_reserve(th_str_t *b, size_t n)
{
    if (b->l + n + 1 <= b->m) return;
    size_t m = b->m ? b->m : 65536;
    while (b->l + n + 1 > m) m += m >> 1;
    b->s = (char *)realloc(b->s, m); b->m = m;
}
static inline void sb_put(th_str_t *b, const char *p, size_t n) { sb_reserve(b, n); memcpy(b->s + b->l, p, n); b->l += n; }
static inline void sb_c(th_str_t *b, char c) { sb_reserve(b, 1); b->s[b->l++] = c; }
static inline void sb_int(th_str_t *b, int v)             /* "%d" */
{
    char tmp[12]; int n = 0; unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    sb_reserve(b, 12);
    if (v < 0) b->s[b->l++] = '-';
    while (n) b->s[b->l++] = tmp[--n];
}
static inline void sb_f1(th_str_t *b, double x) { sb_reserve(b, 340); b->l += (size_t)snprintf(b->s + b->l, 340, "%.1f", x); }     /* "%.1f": the C library's rounding */
static void sb_subpos(th_str_t *b, const int *sp, int pos_n, char term)      /* src/main.c:239-241: +2, the last one +1 */
{
    int i;
    sb_int(b, sp[0] + 2);
    for (i = 1; i < pos_n - 1; ++i) { sb_c(b, ','); sb_int(b, sp[i] + 2); }
    sb_c(b, ','); sb_int(b, sp[i] + 1); sb_c(b, term);
}
static void fmt_read(th_str_t *b, const char *name, const char *seq, int len, th_tandem_seq_t *t, const th_para_t *p)
{
    const size_t nl = strlen(name);
    size_t cs = 0;
    for (int ci = 0; ci < t->cons_n; ++ci) {
        if (p->only_unit) {
            for (int i = 0; i < t->pos_n[ci] - 1; ++i) {
                if (p->out_fmt == TH_FASTA_FMT) {
                    sb_c(b, '>'); sb_put(b, name, nl); sb_put(b, "_rep", 4); sb_int(b, ci); sb_put(b, "_sub", 4); sb_int(b, i); sb_c(b, '\n');
                    sb_put(b, seq + t->sub_pos[ci][i] + 1, (size_t)(t->sub_pos[ci][i + 1] - t->sub_pos[ci][i]));
                    sb_c(b, '\n');
                } else if (p->out_fmt == TH_TAB_FMT) {
                    sb_put(b, name, nl); sb_put(b, "\trep", 4); sb_int(b, ci); sb_put(b, "\tsub", 4); sb_int(b, i); sb_c(b, '\t');
                    const int ul = t->sub_pos[ci][i + 1] - t->sub_pos[ci][i] - 1;           /* '<' at src/main.c:224: one base fewer */
                    if (ul > 0) sb_put(b, seq + t->sub_pos[ci][i] + 1, (size_t)ul);
                    sb_c(b, '\n');
                }
            }
            continue;
        }
        if (p->out_fmt == TH_FASTA_FMT || p->out_fmt == TH_FASTQ_FMT) {      /* ">%s_rep%d_%.1f %d_%d_%d_%d_%.1f_%d_" */
            sb_c(b, p->out_fmt == TH_FASTA_FMT ? '>' : '@'); sb_put(b, name, nl); sb_put(b, "_rep", 4); sb_int(b, ci); sb_c(b, '_'); sb_f1(b, t->copy_num[ci]);
            sb_c(b, ' '); sb_int(b, len); sb_c(b, '_'); sb_int(b, t->cons_start[ci] + 1); sb_c(b, '_'); sb_int(b, t->cons_end[ci] + 1); sb_c(b, '_'); sb_int(b, t->cons_len[ci]);
            sb_c(b, '_'); sb_f1(b, t->ave_match[ci]); sb_c(b, '_'); sb_int(b, t->full_length[ci]); sb_c(b, '_');
            sb_subpos(b, t->sub_pos[ci], t->pos_n[ci], '\n');
        } else {                                                             /* "%s\trep%d\t%.1f\t%d\t%d\t%d\t%d\t%.1f\t%d\t" */
            sb_put(b, name, nl); sb_put(b, "\trep", 4); sb_int(b, ci); sb_c(b, '\t'); sb_f1(b, t->copy_num[ci]);
            sb_c(b, '\t'); sb_int(b, len); sb_c(b, '\t'); sb_int(b, t->cons_start[ci] + 1); sb_c(b, '\t'); sb_int(b, t->cons_end[ci] + 1); sb_c(b, '\t'); sb_int(b, t->cons_len[ci]);
            sb_c(b, '\t'); sb_f1(b, t->ave_match[ci]); sb_c(b, '\t'); sb_int(b, t->full_length[ci]); sb_c(b, '\t');
            sb_subpos(b, t->sub_pos[ci], t->pos_n[ci], '\t');
        }
        sb_put(b, t->cons_seq->seq.s + cs, (size_t)t->cons_len[ci]);
        if (p->out_fmt == TH_FASTQ_FMT) sb_put(b, "\n+\n", 3); else if (p->out_fmt == TH_TAB_QUAL_FMT) sb_c(b, '\t');
        if (p->out_fmt == TH_FASTQ_FMT || p->out_fmt == TH_TAB_QUAL_FMT) sb_put(b, t->cons_seq->qual.s + cs, (size_t)t->cons_len[ci]);
        sb_c(b, '\n');
        cs += (size_t)t->cons_len[ci];
    }
    t->cons_n = 0; t->cons_seq->seq.l = 0; t->cons_seq->qual.l = 0;          /* src/main.c:266-268 (and the quality buffer: DESIGN.md) */
}
/* th_output into memory: what mini_tandem_output() would print for the n reads (slots reset, as there); the buffer is malloc'ed, *len its size */
char *th_format_output(int n, const char *const *names, const char *const *seqs, const int *lens, th_tandem_seq_t *tseq, const th_para_t *p, size_t *len)
{
    th_str_t b = { 0, 0, 0 };
    sb_reserve(&b, 64);
    for (int si = 0; si < n; ++si) fmt_read(&b, names[si], seqs[si], lens[si], tseq + si, p);
    b.s[b.l] = 0;
    *len = b.l;
    return b.s;
}
void th_free_buffer(void *p) { free(p); }

void th_output(FILE *fp, int n, const char *const *names, const char *const *seqs, const int *lens, th_tandem_seq_t *tseq, const th_para_t *p)
{
    th_str_t b = { 0, 0, 0 };
    for (int si = 0; si < n; ++si) {
        fmt_read(&b, names[si], seqs[si], lens[si], tseq + si, p);
        if (b.l >= ((size_t)1 << 20)) { fwrite(b.s, 1, b.l, fp); b.l = 0; }
    }
    if (b.l) fwrite(b.s, 1, b.l, fp);
    free(b.s);
}

/* Record arithmetic of a batch (src/gen_cons.c:208-301 per consensus job): reads are independent and each writes only its
 * own tandem_seq_t slot, so the batch is cut into contiguous read ranges for a few worker threads. */
#define TH_MAX_COLLECT 64
/* text of one batch: buffer t holds the records of the reads collecting thread t took, and the threads' ranges follow each other in read order */
typedef struct { th_str_t buf[TH_MAX_COLLECT]; int n; } th_fmt_t;
typedef struct { th_handle_t *h; const th_batch_t *B; const th_gpu_out_t *o; th_tandem_seq_t *out; int r0, r1, rb; const char *const *names; th_str_t *fmt; double cpu; } collect_arg_t;   /* reads rb + r0 .. rb + r1 of the batch = r0 .. r1 of the part */
static void *collect_range(void *arg)
{
    collect_arg_t *a = (collect_arg_t *)arg;
    const double c0 = thread_cpu_s();
    th_handle_t *h = a->h; const th_gpu_out_t *o = a->o; const th_para_t *p = &h->para; const th_batch_t *B = a->B;
    th_scratch_t sc; memset(&sc, 0, sizeof(sc));
    int lo = 0, hi = o->n_jobs;                  /* first job of read >= r0 (jobs are ordered by read) */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (o->jobs[mid].read < a->r0) lo = mid + 1; else hi = mid; }
    int ji = lo;
    if (a->fmt) a->fmt->l = 0;
    for (int r = a->r0; r < a->r1; ++r) {
        th_tandem_seq_t *t = &a->out[a->rb + r];
        t->cons_n = 0; t->cons_seq->seq.l = 0; t->cons_seq->qual.l = 0;
        while (ji < o->n_jobs && o->jobs[ji].read == r) { finish_job(h, B, o, &o->jobs[ji], t, a->rb, &sc); ++ji; }
        if (B->lens[a->rb + r] >= p->k && p->single_copy == 1 && p->only_full_length && p->five_seq != NULL && p->three_seq != NULL)
            single_copy(B->lens[a->rb + r], B->seqs[a->rb + r], t, p, o->single ? o->single + 40 * r : NULL);   /* src/tidehunter.c:24,49-51 */
        if (a->fmt) fmt_read(a->fmt, a->names[a->rb + r], B->seqs[a->rb + r], B->lens[a->rb + r], t, p);
    }
    scr_free(&sc);
    a->cpu = thread_cpu_s() - c0;
    return NULL;
}
/* Threads one handle may use for the record arithmetic: the CPUs the process may really use (affinity and cgroup quota, not the
 * host's core count) shared out over its live handles, eight at most -- the work is a few hundred milliseconds per million reads and
 * overlaps the kernels of the next batch; -t above the reference's default of 4 raises the cap; TH_HOST_THREADS overrides
 * (bench.py sets it to the rank's share when several processes share a host). */
static int host_threads(const th_handle_t *h)
{
    const int eff = th_effective_cpus();
    int live = __atomic_load_n(&g_live_handles, __ATOMIC_RELAXED); if (live < 1) live = 1;
    int nth = eff / live; if (nth > 8) nth = 8;
    if (h->para.n_thread > 4) nth = h->para.n_thread < eff ? h->para.n_thread : eff;
    const char *e = getenv("TH_HOST_THREADS");
    if (e && atoi(e) > 0) nth = atoi(e);
    return nth > TH_MAX_COLLECT ? TH_MAX_COLLECT : nth < 1 ? 1 : nth;
}
/* names / fmt: NULL, or the reads' names and the batch's text buffers -- then every thread also formats the records of its reads
 * (and resets the slots as th_output would) */
static int batch_collect_s(th_handle_t *h, int out_slot, th_tandem_seq_t *out, const char *const *names, th_fmt_t *fmt)
{
    const th_batch_t *B = &h->res[out_slot & 1];
    th_gpu_out_t o[TH_MAX_SUB];
    for (int s = 0; s < B->n_sub; ++s) th_gpu_get(h->gpu[s], out_slot, &o[s]);
    int nth = host_threads(h);
    if (B->n < 256 * nth) nth = 1;
    collect_arg_t arg[TH_MAX_COLLECT]; pthread_t tid[TH_MAX_COLLECT]; int started[TH_MAX_COLLECT];
    if (nth < B->n_sub) nth = B->n_sub;
    {   /* the threads are shared out over the parts in proportion to their reads, at least one each */
        int first[TH_MAX_SUB + 1]; first[0] = 0;
        for (int sp = 1; sp <= B->n_sub; ++sp) {
            int f = sp == B->n_sub ? nth : (int)((long long)nth * B->sub_off[sp] / (B->n ? B->n : 1));
            if (f < first[sp - 1] + 1) f = first[sp - 1] + 1;
            if (f > nth - (B->n_sub - sp)) f = nth - (B->n_sub - sp);
            first[sp] = f;
        }
        for (int sp = 0; sp < B->n_sub; ++sp)
            for (int t = first[sp]; t < first[sp + 1]; ++t) {
                const int k = t - first[sp], m = first[sp + 1] - first[sp], cnt = B->sub_off[sp + 1] - B->sub_off[sp];
                arg[t].h = h; arg[t].B = B; arg[t].o = &o[sp]; arg[t].out = out; arg[t].rb = B->sub_off[sp];
                arg[t].r0 = (int)((long long)cnt * k / m); arg[t].r1 = (int)((long long)cnt * (k + 1) / m);
                arg[t].names = names; arg[t].fmt = fmt ? &fmt->buf[t] : NULL; arg[t].cpu = 0;
            }
    }
    if (fmt) fmt->n = nth;
    /* a thread that cannot be started does not lose its reads: the range is taken inline */
    for (int t = 1; t < nth; ++t) started[t] = pthread_create(&tid[t], NULL, collect_range, &arg[t]) == 0;
    collect_range(&arg[0]);
    for (int t = 1; t < nth; ++t) { if (started[t]) pthread_join(tid[t], NULL); else collect_range(&arg[t]); }
    double cpu = 0;
    for (int t = 1; t < nth; ++t) if (started[t]) cpu += arg[t].cpu;      /* (thread 0 and inline ranges are on the caller's own clock) */
    cpu_account(CPU_COLLECT, cpu);
    return 0;
}
int th_batch_collect(th_handle_t *h, th_tandem_seq_t *out) { return batch_collect_s(h, 0, out, NULL, NULL); }

int th_process_batch(th_handle_t *h, const char *const *seqs, const int *lens, int n, th_tandem_seq_t *out)
{
    const double t0 = now_s();
    if (th_batch_upload(h, seqs, lens, n)) return -1;
    const double t1 = now_s();
    if (th_batch_run(h)) return -1;
    const double t2 = now_s();
    const int rc = th_batch_collect(h, out);
    if (getenv("TH_TRACE")) {
        double km = 0;
        for (int s = 0; s < h->res[0].n_sub; ++s) { th_gpu_out_t o; th_gpu_get(h->gpu[s], 0, &o); km += o.ms_kernels; }
        fprintf(stderr, "[th_process_batch] %d reads in %d concurrent part(s): upload %.3f s, device run %.3f s (kernels %.3f s summed), collect %.3f s\n", n, h->res[0].n_sub, t1 - t0, t2 - t1, km / 1000.0, now_s() - t2);
    }
    return rc;
}

/* ---------------------------------------------------------------- the batch seam, pipelined */
/* A sequence of batches through one handle as a three-stage software pipeline: while the kernels of batch i run, batch i+1 is
 * uploaded (own copy stream, second input set) and the record arithmetic of batch i-1 fills its tandem_seq_t slots (second
 * result set).  The stages are three threads that hand batches on through counters; the same three stages serve the file
 * driver below (its batches come from the reader instead of an array). */
typedef struct th_stage3 {
    th_handle_t *h;
    pthread_mutex_t mu; pthread_cond_t cv;
    int up_done, run_done, col_done, n_total;       /* batches past each stage; n_total: how many there will be (-1: not known yet) */
    int failed;
    int resident;                /* th_batch_run_steps(): every step runs the batch that sits in input set 0 (nothing is uploaded) */
    double ms_kernel[8];         /* device milliseconds per kernel, summed over the runs */
    /* batch k of this pipeline: reads and result slots (ring of 8: at most 4 batches are in flight) */
    struct { const char *const *seqs; const int *lens; int n; th_tandem_seq_t *out; void *tag; const char *const *names; th_fmt_t *fmt; } b[8];
    void (*on_done)(struct th_stage3 *, int k, void *tag);      /* called by the collecting thread when batch k is complete */
    double busy_up, busy_run, busy_col;
} th_stage3_t;
static void st3_init(th_stage3_t *p, th_handle_t *h) { memset(p, 0, sizeof(*p)); p->h = h; p->n_total = -1; pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->cv, NULL); }
static void st3_destroy(th_stage3_t *p) { pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv); }
static void st3_bump(th_stage3_t *p, int *ctr) { pthread_mutex_lock(&p->mu); ++*ctr; pthread_cond_broadcast(&p->cv); pthread_mutex_unlock(&p->mu); }
static void st3_finish(th_stage3_t *p, int n_total) { pthread_mutex_lock(&p->mu); p->n_total = n_total; pthread_cond_broadcast(&p->cv); pthread_mutex_unlock(&p->mu); }
/* waits until *ctr > k (batch k has passed that stage); returns 0 if the pipeline ends before batch k */
static int st3_wait(th_stage3_t *p, const int *ctr, int k)
{
    int ok = 1;
    pthread_mutex_lock(&p->mu);
    while (*ctr <= k) { if (p->n_total >= 0 && k >= p->n_total) { ok = 0; break; } pthread_cond_wait(&p->cv, &p->mu); }
    pthread_mutex_unlock(&p->mu);
    return ok;
}
/* stage 1, called by whoever produces batches: input set k & 1 is free once batch k-2 has run */
static int st3_upload(th_stage3_t *p, int k, const char *const *seqs, const int *lens, int n, th_tandem_seq_t *out, void *tag, const char *const *names, th_fmt_t *fmt)
{
    if (k >= 2) st3_wait(p, &p->run_done, k - 2);
    const double t0 = now_s();
    p->b[k & 7].seqs = seqs; p->b[k & 7].lens = lens; p->b[k & 7].n = n; p->b[k & 7].out = out; p->b[k & 7].tag = tag; p->b[k & 7].names = names; p->b[k & 7].fmt = fmt;
    if (!p->failed && n > 0 && batch_upload_s(p->h, k, seqs, lens, n, 0)) p->failed = 1;
    p->busy_up += now_s() - t0;
    st3_bump(p, &p->up_done);
    return p->failed ? -1 : 0;
}
static void *st3_runner(void *arg)
{
    th_stage3_t *p = (th_stage3_t *)arg;
    for (int k = 0;; ++k) {
        if (!st3_wait(p, &p->up_done, k)) { cpu_account(CPU_RUN, thread_cpu_s()); return NULL; }
        if (k >= 2) st3_wait(p, &p->col_done, k - 2);            /* result set k & 1 is free once batch k-2 has been collected */
        const double t0 = now_s();
        if (!p->failed && p->b[k & 7].n > 0 && batch_run_s(p->h, p->resident ? 0 : k, k)) p->failed = 1;
        if (!p->failed && p->b[k & 7].n > 0)
            for (int s = 0; s < p->h->res[k & 1].n_sub; ++s) { th_gpu_out_t o; th_gpu_get(p->h->gpu[s], k, &o); for (int i = 0; i < 8; ++i) p->ms_kernel[i] += o.ms_kernel[i]; }
        p->busy_run += now_s() - t0;
        st3_bump(p, &p->run_done);
    }
}
static void *st3_collector(void *arg)
{
    th_stage3_t *p = (th_stage3_t *)arg;
    for (int k = 0;; ++k) {
        if (!st3_wait(p, &p->run_done, k)) { cpu_account(CPU_COLLECT, thread_cpu_s()); return NULL; }
        const double t0 = now_s();
        if (p->b[k & 7].fmt) p->b[k & 7].fmt->n = 0;
        if (!p->failed && p->b[k & 7].n > 0 && batch_collect_s(p->h, k, p->b[k & 7].out, p->b[k & 7].names, p->b[k & 7].fmt)) p->failed = 1;
        p->busy_col += now_s() - t0;
        if (p->on_done) p->on_done(p, k, p->b[k & 7].tag);
        st3_bump(p, &p->col_done);
    }
}
int th_process_batches(th_handle_t *h, int n_batches, const char *const *const *seqs, const int *const *lens, const int *n, th_tandem_seq_t *const *out)
{
    th_stage3_t p; st3_init(&p, h);
    pthread_t t_run, t_col;
    if (pthread_create(&t_run, NULL, st3_runner, &p) != 0) { st3_destroy(&p); return host_fail("pthread_create failed"); }
    if (pthread_create(&t_col, NULL, st3_collector, &p) != 0) { st3_finish(&p, 0); pthread_join(t_run, NULL); st3_destroy(&p); return host_fail("pthread_create failed"); }
    for (int k = 0; k < n_batches; ++k) st3_upload(&p, k, seqs[k], lens[k], n[k], out[k], NULL, NULL, NULL);
    st3_finish(&p, n_batches);
    pthread_join(t_run, NULL); pthread_join(t_col, NULL);
    const int bad = p.failed;
    if (getenv("TH_TRACE")) fprintf(stderr, "[th_process_batches] %d batches: busy upload %.3f s, run %.3f s, collect %.3f s\n", n_batches, p.busy_up, p.busy_run, p.busy_col);
    st3_destroy(&p);
    if (bad && !g_err[0]) snprintf(g_err, sizeof(g_err), "a batch failed on the device (see stderr)");
    return bad ? -1 : 0;
}

/* The resident batch (th_batch_upload) run `steps` times: the kernels of pass k+1 overlap the record arithmetic of pass k (second
 * result set), as in the file driver; every pass fills out[0..n).  ms_kernel (8 doubles, optional) receives the device milliseconds
 * per kernel summed over the passes.  This is bench.py's timed region: inputs resident, K passes, nothing else inside. */
int th_batch_run_steps(th_handle_t *h, int steps, th_tandem_seq_t *out, double *ms_kernel)
{
    th_stage3_t p; st3_init(&p, h); p.resident = 1;
    pthread_t t_run, t_col;
    if (pthread_create(&t_run, NULL, st3_runner, &p) != 0) { st3_destroy(&p); return host_fail("pthread_create failed"); }
    if (pthread_create(&t_col, NULL, st3_collector, &p) != 0) { st3_finish(&p, 0); pthread_join(t_run, NULL); st3_destroy(&p); return host_fail("pthread_create failed"); }
    for (int k = 0; k < steps; ++k) {
        if (k >= 2) st3_wait(&p, &p.run_done, k - 2);
        p.b[k & 7].seqs = h->up[0].seqs; p.b[k & 7].lens = h->up[0].lens; p.b[k & 7].n = h->up[0].n; p.b[k & 7].out = out; p.b[k & 7].tag = NULL; p.b[k & 7].names = NULL; p.b[k & 7].fmt = NULL;
        st3_bump(&p, &p.up_done);
    }
    st3_finish(&p, steps);
    pthread_join(t_run, NULL); pthread_join(t_col, NULL);
    const int bad = p.failed;
    if (ms_kernel) for (int i = 0; i < 8; ++i) ms_kernel[i] = p.ms_kernel[i];
    st3_destroy(&p);
    if (bad && !g_err[0]) snprintf(g_err, sizeof(g_err), "a pass failed on the device (see stderr)");
    return bad ? -1 : 0;
}

/* ---------------------------------------------------------------- whole-file driver, src/main.c:375-436 */
/* th_run_file = mini_tandem() (src/main.c:377-447).  The reference alternates "read a chunk / fan out / print"; here
 * reader -> upload -> kernels -> record arithmetic -> writer run as a pipeline over a ring of batch slots, so that parsing and
 * uploading batch i+1 and collecting / formatting batch i-1 overlap the device work on batch i.  Output
 * order = input order (src/main.c:217). */
enum { SLOT_EMPTY = 0, SLOT_FILLED = 1, SLOT_DONE = 2 };
#define TH_MAX_DEV 16
#define TH_MAX_WORKER (4 * TH_MAX_DEV)          /* up to four pipelines (handles) per device */
#define TH_MAX_SLOT (5 * TH_MAX_WORKER + 2)
typedef struct { char **names, **seqs; int *lens; int n, state, batch; th_tandem_seq_t *ts; th_str_t arena; size_t *noff, *soff; th_fmt_t fmt; } th_slot_t;   /* batch: index of the batch the slot holds */
typedef struct {
    th_slot_t slot[TH_MAX_SLOT]; int n_slot, n_dev, n_primary;     /* n_primary: listed devices; pipelines n_primary.. are the further ones per device */
    pthread_mutex_t mu; pthread_cond_t cv;
    th_reader_t *rd; FILE *out; th_para_t *p; int batch_reads, ramp0; volatile int stop;
    int eof_batch;               /* index of the empty batch that ends the input (-1 until the reader meets it) */
    int next_batch;              /* ticket counter of the device pipelines (the reference's THREAD_READ_I, src/main.c:277-281, at batch granularity) */
    int n_reserved;              /* pipelines whose device buffers are in place (none launches a kernel before all are) */
    double t_read, t_write;      /* busy seconds of the two side threads (TH_TRACE=1 prints them) */
    double t_start; int trace;
} th_pipe_t;
/* steady-state bookkeeping per pipeline: when its first full-size batch was complete, how many of its reads were complete by then, and
 * when its last batch was complete -- the rate over that window excludes start-up (HIP initialisation, device allocation, the ramp) and,
 * taken per pipeline, is not inflated by another pipeline's batch that happens to complete just after the window opens */
typedef struct { th_pipe_t *q; int dev_index, device; int rc, total, batches; th_stage3_t st; double t_first_full, t_last; long long reads_done, reads_at_first_full; } th_dev_worker_t;

/* waits until the slot holds batch `batch` in state `state` (batch < 0: any occupant -- the reader waiting for a free slot);
 * returns 0 if that batch lies beyond the end of the input instead */
static int slot_wait(th_pipe_t *q, th_slot_t *s, int state, int batch)
{
    int ok = 1;
    pthread_mutex_lock(&q->mu);
    while (!(s->state == state && (batch < 0 || s->batch == batch))) {
        if (batch >= 0 && q->eof_batch >= 0 && batch > q->eof_batch) { ok = 0; break; }
        pthread_cond_wait(&q->cv, &q->mu);
    }
    pthread_mutex_unlock(&q->mu);
    return ok;
}
static void slot_set(th_pipe_t *q, th_slot_t *s, int state)
{
    pthread_mutex_lock(&q->mu); s->state = state; pthread_cond_broadcast(&q->cv); pthread_mutex_unlock(&q->mu);
}
static void *pipe_reader(void *arg)
{
    th_pipe_t *q = (th_pipe_t *)arg;
    const long long max_bases = 1LL << 30;
    for (int i = 0;; ++i) {
        th_slot_t *s = &q->slot[i % q->n_slot];
        slot_wait(q, s, SLOT_EMPTY, -1);
        int n = 0; long long bases = 0;
        const double t0 = now_s();
        s->arena.l = 0;
        /* the first batches are small and grow fourfold (ramp0, 4 ramp0, ... up to batch_reads): the device starts after a few
         * milliseconds of parsing instead of after a whole batch, and the pipeline fills while the batches grow */
        const long long target = i < 10 && ((long long)q->ramp0 << (2 * i)) < q->batch_reads ? ((long long)q->ramp0 << (2 * i)) : q->batch_reads;
        if (!q->stop)
            while (n < target && bases < max_bases && rd_next_into(q->rd, &s->arena, &s->noff[n], &s->soff[n], (const char **)&s->seqs[n], &s->lens[n])) { bases += s->lens[n]; ++n; }
        for (int k = 0; k < n; ++k) { s->names[k] = s->arena.s + s->noff[k]; if (s->soff[k] != (size_t)-1) s->seqs[k] = s->arena.s + s->soff[k]; }
        q->t_read += now_s() - t0;
        s->n = n; s->batch = i;
        if (n == 0) { pthread_mutex_lock(&q->mu); q->eof_batch = i; pthread_mutex_unlock(&q->mu); }
        slot_set(q, s, SLOT_FILLED);
        if (n == 0) { cpu_account(CPU_READER, thread_cpu_s()); return NULL; }
    }
}
static void *pipe_writer(void *arg)
{
    th_pipe_t *q = (th_pipe_t *)arg;
    for (int i = 0;; ++i) {
        th_slot_t *s = &q->slot[i % q->n_slot];
        slot_wait(q, s, SLOT_DONE, i);
        const int n = s->n;
        const double t0 = now_s();
        /* the text was formatted by the batch's collecting threads, one buffer per thread in read order: nothing per record happens here */
        if (n > 0 && q->out && !q->stop) for (int t = 0; t < s->fmt.n; ++t) if (s->fmt.buf[t].l) fwrite(s->fmt.buf[t].s, 1, s->fmt.buf[t].l, q->out);
        s->fmt.n = 0;
        q->t_write += now_s() - t0;
        slot_set(q, s, SLOT_EMPTY);
        if (n == 0) { cpu_account(CPU_WRITER, thread_cpu_s()); return NULL; }
    }
}
/* the collecting thread of a device pipeline has finished a batch: its slot goes to the writer */
static void pipe_batch_done(th_stage3_t *st, int k, void *tag)
{
    th_dev_worker_t *w = (th_dev_worker_t *)((char *)st - offsetof(th_dev_worker_t, st)); th_pipe_t *q = w->q; th_slot_t *s = (th_slot_t *)tag;
    (void)k;
    if (st->failed) { w->rc = -1; q->stop = 1; } else w->total += s->n;     /* never partial output: a failed batch and everything after it are dropped */
    const double t = now_s();
    w->reads_done += s->n; if (s->n > 0) w->t_last = t;
    if (w->t_first_full == 0 && s->n >= q->batch_reads) { w->t_first_full = t; w->reads_at_first_full = w->reads_done; }
    if (q->trace) fprintf(stderr, "[th_run_file]   pipeline %d: batch %d (%d reads) complete at %.3f s\n", w->dev_index, s->batch, s->n, t - q->t_start);
    slot_set(q, s, SLOT_DONE);
}
/* One pipeline per device: this thread creates the handle (HIP initialisation overlaps the reader's first batch), takes
 * batches from the shared ticket counter as the reference's threads take reads (src/main.c:273-291) -- a device that finishes
 * early (shorter reads, shorter periods) simply takes more batches, stealing with no extra mechanism -- and uploads them; two
 * more threads run the kernels and do the record arithmetic (th_stage3_t).  Reads are independent, so devices exchange
 * nothing; the writer restores input order.  A pipeline holds at most four batches (uploaded, running, being collected, being
 * written) and the ring has five slots per pipeline plus the reader's. */
static void *pipe_device(void *arg)
{
    th_dev_worker_t *w = (th_dev_worker_t *)arg; th_pipe_t *q = w->q;
    int fsub = 1;
    { const char *e = getenv("TH_FILE_SUB"); if (e && atoi(e) >= 1 && atoi(e) <= TH_MAX_SUB) fsub = atoi(e); }
    th_handle_t *h = th_init_sub(q->p, w->device, fsub);
    if (!h) { w->rc = -1; q->stop = 1; }
    if (q->trace) fprintf(stderr, "[th_run_file]   pipeline %d: handle ready at %.3f s\n", w->dev_index, now_s() - q->t_start);
    /* Every pipeline sizes its device buffers for full batches from the first parsed batch (peeked, not taken) before any pipeline
     * launches a kernel: an allocation issued next to another pipeline's running kernels waits for a kernel boundary -- seconds at
     * these sizes -- while on an idle device the allocations take milliseconds.  The first pipeline of each device goes first; a
     * further one joins only if its buffers fit in a third of what is then free, and retires otherwise. */
    int retired = 0;
    {
        th_slot_t *s0 = &q->slot[0];
        const int have = slot_wait(q, s0, SLOT_FILLED, 0) && s0->n > 0;
        const int round = w->dev_index / q->n_primary;
        pthread_mutex_lock(&q->mu);
        while (q->n_reserved < round * q->n_primary && !q->stop) pthread_cond_wait(&q->cv, &q->mu);
        pthread_mutex_unlock(&q->mu);
        if (have && h && !q->stop) {
            long long bases = 0; for (int x = 0; x < s0->n; ++x) bases += s0->lens[x];
            const long long full = (long long)((double)bases / s0->n * q->batch_reads * 1.1) + 65536;
            if (round > 0 && (s0->n < q->ramp0 || !th_gpu_room_for(h->gpu[0], q->batch_reads, full))) retired = 1;     /* (an input smaller than the first ramp step needs one pipeline) */
            else if (s0->n < q->batch_reads) th_batch_reserve(h, q->batch_reads, full);
        }
        pthread_mutex_lock(&q->mu);
        ++q->n_reserved; pthread_cond_broadcast(&q->cv);
        while (q->n_reserved < q->n_dev && !q->stop) pthread_cond_wait(&q->cv, &q->mu);
        pthread_mutex_unlock(&q->mu);
        if (q->trace) fprintf(stderr, "[th_run_file]   pipeline %d: %s at %.3f s\n", w->dev_index, retired ? "retired (one pipeline is enough for this input / device memory)" : "buffers reserved", now_s() - q->t_start);
    }
    if (retired) { st3_init(&w->st, NULL); if (h) th_destroy(h); cpu_account(CPU_UPLOAD, thread_cpu_s()); return NULL; }
    st3_init(&w->st, h); w->st.on_done = pipe_batch_done; w->st.failed = h ? 0 : 1;
    pthread_t t_run, t_col;
    int have_threads = pthread_create(&t_run, NULL, st3_runner, &w->st) == 0;
    if (have_threads && pthread_create(&t_col, NULL, st3_collector, &w->st) != 0) { st3_finish(&w->st, 0); pthread_join(t_run, NULL); have_threads = 0; }
    if (!have_threads) {
        /* no pipeline without its threads: the run fails, and the batches this pipeline draws go to the writer untouched (it prints
         * nothing once `stop` is set) so that reader, writer and the other pipelines still come to an end */
        host_fail("pthread_create failed"); w->rc = -1; q->stop = 1;
        for (;;) {
            const int i = __atomic_fetch_add(&q->next_batch, 1, __ATOMIC_RELAXED);
            th_slot_t *s = &q->slot[i % q->n_slot];
            if (!slot_wait(q, s, SLOT_FILLED, i)) break;
            const int last = s->n == 0;
            s->fmt.n = 0; slot_set(q, s, SLOT_DONE);
            if (last) break;
        }
        if (h) th_destroy(h);
        return NULL;
    }
    int k = 0;
    for (;; ++k) {
        if (k >= 2) st3_wait(&w->st, &w->st.run_done, k - 2);       /* take a ticket only when an input set is free: an idle device gets the batch instead */
        const int i = __atomic_fetch_add(&q->next_batch, 1, __ATOMIC_RELAXED);
        th_slot_t *s = &q->slot[i % q->n_slot];
        if (!slot_wait(q, s, SLOT_FILLED, i)) break;
        if (q->stop) w->st.failed = 1;
        st3_upload(&w->st, k, (const char *const *)s->seqs, s->lens, s->n, s->ts, s, (const char *const *)s->names, &s->fmt);
        ++w->batches;
        if (s->n == 0) { ++k; break; }                              /* the empty batch that ends the input went through (the writer stops on it) */
    }
    st3_finish(&w->st, k);
    pthread_join(t_run, NULL); pthread_join(t_col, NULL);
    if (w->st.failed) w->rc = -1;
    if (h) th_destroy(h);
    cpu_account(CPU_UPLOAD, thread_cpu_s());
    return NULL;
}

int th_run_file_multi(const char *read_fn, th_para_t *p, FILE *out, const int *devices_in, int n_dev_in, int batch_reads)
{
    if (n_dev_in < 1 || n_dev_in > TH_MAX_DEV) return host_fail("between 1 and 16 devices");
    /* 65536 reads (and at most 2^30 bases, pipe_reader) per batch since round 5: per-position arrays + POA scratch of one pipeline ~70 GB of HBM at 10 kb reads, so the second
     * pipeline of a device retires (pipe_device: an arena must fit a third of the free memory) and ONE pipeline of 65536-read batches runs.  The matrix on 1 000 000 headline
     * reads after the round's kernel work (gpurun_out e2e sweep, round 5): 2 x 32768 (the default until then) 146.6 k reads/s steady, 1 x 65536 154.4 k, 2 x 65536 153.5 k,
     * 1 x 100000 153.7 k, 1 x 50000 151.0 k, 3 x 32768 148.3 k -- with the kernels 10 % faster the half-size batches' grid-fill granularity (5 POA jobs per wave instead of 10)
     * costs more than the overlap of two pipelines hides. */
    if (batch_reads <= 0) batch_reads = 65536;
    /* Two pipelines (handle + upload / run / collect threads each) per listed device: while one is inside its POA / alignment
     * launches the other's short, latency-bound kernels (hits, sort, extraction, partition) and its three synchronisations are
     * hidden -- 116.2 k against 112.4 k reads/s on 1 000 000 reads of config 3 in round 3 (two pipelines of 32768-read batches against one of
     * 65536: the same memory; profiles/r03_e2e_pipelines.txt; with `-b 32768` that is still what runs).  TH_WORKERS_PER_DEV=1..4 overrides; a second pipeline
     * that does not fit the device's memory retires (pipe_device). */
    int wpd = 2;
    { const char *e = getenv("TH_WORKERS_PER_DEV"); if (e && atoi(e) >= 1 && atoi(e) <= 4) wpd = atoi(e); }
    int devices[TH_MAX_WORKER], n_dev = 0;
    for (int r = 0; r < wpd; ++r) for (int d = 0; d < n_dev_in; ++d) devices[n_dev++] = devices_in[d];
    if (p->five_fn && p->three_fn && !p->five_seq) if (th_load_adapters(p)) return -1;
    th_reader_t *r = rd_open(read_fn);
    if (!r) { char m[300]; snprintf(m, sizeof(m), "fail to open file '%s'", read_fn); return host_fail(m); }
    th_pipe_t *q = (th_pipe_t *)calloc(1, sizeof(th_pipe_t));
    pthread_mutex_init(&q->mu, NULL); pthread_cond_init(&q->cv, NULL);
    q->rd = r; q->out = out; q->p = p; q->batch_reads = batch_reads; q->n_dev = n_dev; q->n_primary = n_dev_in; q->n_slot = 5 * n_dev + 1; q->eof_batch = -1;
    q->ramp0 = batch_reads / 4 > 2048 ? batch_reads / 4 : 2048;       /* one small batch, then full ones: a mapped file parses at ~10 GB/s, so only the first launch has to wait */
    { const char *e = getenv("TH_RAMP0"); if (e && atoi(e) > 0) q->ramp0 = atoi(e); }
    for (int i = 0; i < q->n_slot; ++i) {
        q->slot[i].names = (char **)calloc(batch_reads, sizeof(char *)); q->slot[i].seqs = (char **)calloc(batch_reads, sizeof(char *));
        q->slot[i].lens = (int *)calloc(batch_reads, sizeof(int)); q->slot[i].ts = th_alloc_tandem_seq(batch_reads);
        q->slot[i].noff = (size_t *)calloc(batch_reads, sizeof(size_t)); q->slot[i].soff = (size_t *)calloc(batch_reads, sizeof(size_t));
    }
    pthread_t t_rd, t_wr, t_dev[TH_MAX_WORKER]; th_dev_worker_t *w = (th_dev_worker_t *)calloc(TH_MAX_WORKER, sizeof(th_dev_worker_t));
    const double t_start = now_s();
    q->t_start = t_start; q->trace = getenv("TH_TRACE") != NULL;
    double cpu0[CPU_N]; th_host_cpu_seconds(cpu0, CPU_N);
    struct rusage ru0; getrusage(RUSAGE_SELF, &ru0);
    int total = 0, rc = 0; double busy_up = 0, busy_run = 0, busy_col = 0;
    int dev_on[TH_MAX_WORKER]; memset(dev_on, 0, sizeof(dev_on));
    /* the writer first, then the reader: whichever cannot be started, the other is not left waiting */
    const int wr_on = pthread_create(&t_wr, NULL, pipe_writer, q) == 0;
    int rd_on = 0;
    if (wr_on && !(rd_on = pthread_create(&t_rd, NULL, pipe_reader, q) == 0)) {
        q->stop = 1; q->slot[0].n = 0; q->slot[0].batch = 0; q->slot[0].fmt.n = 0;
        pthread_mutex_lock(&q->mu); q->eof_batch = 0; pthread_mutex_unlock(&q->mu);
        slot_set(q, &q->slot[0], SLOT_DONE);          /* the empty batch that ends the writer */
    }
    if (!rd_on || !wr_on) { rc = -1; host_fail("pthread_create failed"); }
    int n_started = 0;
    if (rd_on && wr_on)
        for (int d = 0; d < n_dev; ++d) {
            w[d].q = q; w[d].dev_index = d; w[d].device = devices[d];
            dev_on[d] = pthread_create(&t_dev[d], NULL, pipe_device, &w[d]) == 0;
            n_started += dev_on[d];
        }
    if (rd_on && wr_on && n_started < n_dev) {
        /* a pipeline thread that could not be started: the pipelines that did start wait for n_dev reservations, so the run is stopped
         * (which also ends those waits) and fails; with no pipeline at all the calling thread draws the batches, as a pipeline without
         * threads does (pipe_device), so that reader and writer come to an end */
        rc = -1; host_fail("pthread_create failed");
        pthread_mutex_lock(&q->mu); q->stop = 1; q->n_reserved += n_dev - n_started; pthread_cond_broadcast(&q->cv); pthread_mutex_unlock(&q->mu);
        if (n_started == 0)
            for (;;) {
                const int i = __atomic_fetch_add(&q->next_batch, 1, __ATOMIC_RELAXED);
                th_slot_t *sl = &q->slot[i % q->n_slot];
                if (!slot_wait(q, sl, SLOT_FILLED, i)) break;
                const int last = sl->n == 0;
                sl->fmt.n = 0; slot_set(q, sl, SLOT_DONE);
                if (last) break;
            }
    }
    for (int d = 0; d < n_dev; ++d) if (dev_on[d]) { pthread_join(t_dev[d], NULL); total += w[d].total; rc |= w[d].rc; busy_up += w[d].st.busy_up; busy_run += w[d].st.busy_run; busy_col += w[d].st.busy_col; }
    if (rd_on) pthread_join(t_rd, NULL);
    if (wr_on) pthread_join(t_wr, NULL);
    if (getenv("TH_TRACE")) {
        const double t_end = now_s();
        /* steady: per pipeline, the reads completed after its first full-size batch over the time from then to its last batch; the
         * pipelines' rates add up.  0 when the input is too short for a window. */
        double steady = 0, window = 0;
        for (int d = 0; d < n_dev; ++d)
            if (w[d].t_first_full > 0 && w[d].t_last > w[d].t_first_full && w[d].reads_done > w[d].reads_at_first_full) {
                steady += (w[d].reads_done - w[d].reads_at_first_full) / (w[d].t_last - w[d].t_first_full);
                if (w[d].t_last - w[d].t_first_full > window) window = w[d].t_last - w[d].t_first_full;
            }
        fprintf(stderr, "[th_run_file] %d reads on %d device pipeline(s): wall %.3f s; busy: reader %.3f s, upload %.3f s, device runs %.3f s, record arithmetic %.3f s, writer %.3f s; steady %.1f reads/s over the last %.3f s\n",
                total, n_dev, t_end - t_start, q->t_read, busy_up, busy_run, busy_col, q->t_write, steady, window);
        if (n_dev > 1) for (int d = 0; d < n_dev; ++d) fprintf(stderr, "[th_run_file]   pipeline %d (device %d): %d batches, %d reads, device busy %.3f s\n", d, devices[d], w[d].batches, w[d].total, w[d].st.busy_run);
        /* host CPU seconds by role (thread CPU clocks) and in all (getrusage); the rest is the HIP runtime's own threads */
        double cpu1[CPU_N], sum = 0; th_host_cpu_seconds(cpu1, CPU_N);
        struct rusage ru1; getrusage(RUSAGE_SELF, &ru1);
        for (int i = 0; i < CPU_N; ++i) { cpu1[i] -= cpu0[i]; sum += cpu1[i]; }
        const double all = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + 1e-6 * ((ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec) + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec));
        fprintf(stderr, "[th_run_file] host CPU %.3f s (%.2f per 10^6 reads; %d usable CPUs): reader %.3f, inflate %.3f, upload threads %.3f, run threads %.3f, record arithmetic + formatting %.3f, writer %.3f, runtime and the rest %.3f\n",
                all, total > 0 ? all * 1e6 / total : 0.0, th_effective_cpus(), cpu1[CPU_READER], cpu1[CPU_INFLATE], cpu1[CPU_UPLOAD], cpu1[CPU_RUN], cpu1[CPU_COLLECT], cpu1[CPU_WRITER], all - sum);
    }
    for (int d = 0; d < n_dev; ++d) st3_destroy(&w[d].st);
    for (int i = 0; i < q->n_slot; ++i) { th_free_tandem_seq(q->slot[i].ts, batch_reads); free(q->slot[i].names); free(q->slot[i].seqs); free(q->slot[i].lens); free(q->slot[i].noff); free(q->slot[i].soff); free(q->slot[i].arena.s); for (int t = 0; t < TH_MAX_COLLECT; ++t) free(q->slot[i].fmt.buf[t].s); }
    pthread_mutex_destroy(&q->mu); pthread_cond_destroy(&q->cv);
    if (r->failed && !rc) { rc = -1; host_fail("the input is corrupt or truncated: the output holds only the reads before the damage"); }
    free(q); free(w); rd_close(r);
    return rc ? -1 : total;
}
int th_run_file(const char *read_fn, th_para_t *p, FILE *out, int device, int batch_reads)
{
    return th_run_file_multi(read_fn, p, out, &device, 1, batch_reads);
}

/* ---------------------------------------------------------------- the reader alone */
/* Parses read_fn with the reader th_run_file uses, batch slot and all, without touching a device: the record count is returned,
 * *n_bases receives the bases, *digest (optional: it costs about as much as the parsing) an FNV-1a hash over every name and
 * sequence in order, so that the mmap / BGZF / stream paths can be compared.  -1 if the file cannot be opened. */
long long th_parse_file(const char *read_fn, long long *n_bases, unsigned long long *digest)
{
    th_reader_t *r = rd_open(read_fn);
    if (!r) return -1;
    th_str_t a = { 0, 0, 0 };
    long long n = 0, bases = 0; unsigned long long h = 1469598103934665603ull;
    for (;;) {
        a.l = 0;
        int k = 0, len; size_t no, so; const char *zc;
        while (k < 4096 && rd_next_into(r, &a, &no, &so, &zc, &len)) {
            if (digest) {
                const unsigned char *nm = (const unsigned char *)a.s + no;
                for (; *nm; ++nm) h = (h ^ *nm) * 1099511628211ull;
                const unsigned char *sq = (const unsigned char *)(so == (size_t)-1 ? zc : a.s + so);
                unsigned long long h2 = h;
                for (int i = 0; i < len; ++i) h2 = (h2 ^ sq[i]) * 1099511628211ull;
                h = h2 ^ (unsigned long long)len;
            }
            bases += len; ++k;
        }
        n += k;
        if (k < 4096) break;
    }
    const int damaged = r->failed;
    free(a.s); rd_close(r);
    if (damaged) return -2;                       /* corrupt or truncated input */
    if (n_bases) *n_bases = bases;
    if (digest) *digest = h;
    return n;
}

/* ---------------------------------------------------------------- synthetic reads */
static uint64_t sm64(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double sm_u(uint64_t *s) { return (double)(sm64(s) >> 11) / 9007199254740992.0; }
int th_synth_read(uint64_t seed, uint64_t read_index, int period, int copies, double err, char *dst, int cap)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + read_index * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull;
    char *unit = (char *)malloc((size_t)period);
    for (int i = 0; i < period; ++i) unit[i] = "ACGT"[sm64(&s) >> 62];
    int n = 0;
    for (int c = 0; c < copies; ++c)
        for (int i = 0; i < period; ++i) {
            char b = unit[i];
            if (sm_u(&s) < err) {
                int kind = (int)(sm_u(&s) * 3);
                if (kind == 0) { int bi = (int)(strchr("ACGT", b) - "ACGT"); b = "ACGT"[(bi + 1 + (int)(sm_u(&s) * 3)) & 3]; if (n < cap) dst[n] = b; ++n; }
                else if (kind == 1) { if (n < cap) dst[n] = b; ++n; char x = "ACGT"[(int)(sm_u(&s) * 4) & 3]; if (n < cap) dst[n] = x; ++n; }
                /* kind 2: deletion */
            } else { if (n < cap) dst[n] = b; ++n; }
        }
    free(unit);
    return n;
}

/* read r has period {200, 500, 1000, 2000, 5000}[r mod 5], copies = max(3, 10000 / period); the unit
 * is 5' adapter + random insert + reverse complement of the 3' adapter, copied with the same per-base edit model as
 * th_synth_read(); every second group of five reads is reverse-complemented (both strands).  Adapters come from
 * th_synth_adapter(seed 55, ...) in the configs' definition; any pair can be passed. */
void th_synth_adapter(uint64_t seed, int which, char *dst, int len)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(which + 1) * 0xA24BAED4963EE407ull;
    for (int i = 0; i < len; ++i) dst[i] = "ACGT"[sm64(&s) >> 62];
    dst[len] = 0;
}
int th_synth_read_c5(uint64_t seed, uint64_t read_index, const char *five, int five_len, const char *three, int three_len, double err, char *dst, int cap)
{
    static const int periods[5] = { 200, 500, 1000, 2000, 5000 };
    const int period = periods[read_index % 5], copies = 10000 / period > 3 ? 10000 / period : 3;
    if (five_len + three_len >= period) return -1;
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + read_index * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull;
    char *unit = (char *)malloc((size_t)period), *trc = rc_seq(three, three_len);
    memcpy(unit, five, (size_t)five_len);
    for (int i = five_len; i < period - three_len; ++i) unit[i] = "ACGT"[sm64(&s) >> 62];
    memcpy(unit + period - three_len, trc, (size_t)three_len);
    free(trc);
    int n = 0;
    for (int c = 0; c < copies; ++c)
        for (int i = 0; i < period; ++i) {
            char b = unit[i];
            if (sm_u(&s) < err) {
                int kind = (int)(sm_u(&s) * 3);
                if (kind == 0) { int bi = (int)(strchr("ACGT", b) - "ACGT"); b = "ACGT"[(bi + 1 + (int)(sm_u(&s) * 3)) & 3]; if (n < cap) dst[n] = b; ++n; }
                else if (kind == 1) { if (n < cap) dst[n] = b; ++n; char x = "ACGT"[(int)(sm_u(&s) * 4) & 3]; if (n < cap) dst[n] = x; ++n; }
            } else { if (n < cap) dst[n] = b; ++n; }
        }
    free(unit);
    if (((read_index / 5) & 1) && n <= cap) {     /* antisense strand */
        char *r = rc_seq(dst, n);
        memcpy(dst, r, (size_t)n); free(r);
    }
    return n;
}

/* ---------------------------------------------------------------- shared ticket counters (multi-process work stealing) */
/* The reference balances its threads with one shared counter (THREAD_READ_I under a write lock, src/main.c:208-210,277-281).
 * With one process per GPU the counter has to live in memory all ranks of the node can reach: a page of POSIX shared memory,
 * 512 independent 64-bit counters, advanced with an atomic fetch-add (no lock, no collective). */
#define TH_TICKET_N 512
struct th_ticket { long long *c; char name[128]; };
th_ticket_t *th_ticket_open(const char *name, int create)
{
    /* the creator starts from a fresh, zero-filled segment: a run that was killed may have left one of this name behind with its
     * counters advanced (reopening it made the next run skip blocks) */
    if (create) shm_unlink(name);
    int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) { host_fail("shm_open failed for the ticket counters"); return NULL; }
    if (create && ftruncate(fd, TH_TICKET_N * (off_t)sizeof(long long)) != 0) { close(fd); host_fail("ftruncate failed for the ticket counters"); return NULL; }
    void *m = mmap(NULL, TH_TICKET_N * sizeof(long long), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { host_fail("mmap failed for the ticket counters"); return NULL; }
    if (create) memset(m, 0, TH_TICKET_N * sizeof(long long));
    th_ticket_t *t = (th_ticket_t *)calloc(1, sizeof(*t));
    t->c = (long long *)m; snprintf(t->name, sizeof(t->name), "%s", name);
    return t;
}
long long th_ticket_next(th_ticket_t *t, int counter)
{
    if (!t || counter < 0 || counter >= TH_TICKET_N) return -1;
    return __atomic_fetch_add(&t->c[counter], 1LL, __ATOMIC_RELAXED);
}
void th_ticket_close(th_ticket_t *t, int unlink_it)
{
    if (!t) return;
    munmap(t->c, TH_TICKET_N * sizeof(long long));
    if (unlink_it) shm_unlink(t->name);
    free(t);
}
