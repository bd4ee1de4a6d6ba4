/*
 * oracle/tho_io.c -- TEST INFRASTRUCTURE (see tho.h).
 * Parameters, FASTA/FASTQ(.gz) reading, record formatting and the whole-file driver.
 * Restates src/main.c:173-182,214-271,273-291,329-364,375-436 and the kseq record grammar
 * (src/kseq.h:176-215: name = first word of the header line, sequence lines concatenated,
 * '+' starts qualities of the same length).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <ctype.h>
#include <pthread.h>
#include <zlib.h>
#include "tho.h"

void tho_para_init(tho_para_t *p)
{
    memset(p, 0, sizeof(*p));
    p->n_thread = 4; p->ada_match_rat = 0.8f;
    p->k = 8; p->w = 1; p->hpc = 0; p->min_copy = 2; p->max_div = 0.25;
    p->min_p = 30; p->max_p = 10000; p->min_cov = 0; p->min_frac = 0.0;
    p->match = 2; p->mismatch = 4; p->gap_open1 = 4; p->gap_open2 = 24; p->gap_ext1 = 2; p->gap_ext2 = 1;
    p->min_len = 30; p->out_fmt = THO_FASTA_FMT;
}

/* ---- reader ---- */
struct tho_reader { gzFile fp; unsigned char *buf; int beg, end, eof; int last; };

tho_reader_t *tho_reader_open(const char *fn)
{
    gzFile fp = strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(0, "r");
    if (!fp) return NULL;
    tho_reader_t *r = (tho_reader_t *)calloc(1, sizeof(*r));
    r->fp = fp; r->buf = (unsigned char *)malloc(1 << 16);
    return r;
}
void tho_reader_close(tho_reader_t *r) { if (r) { gzclose(r->fp); free(r->buf); free(r); } }
void tho_read_free(tho_read_t *r) { free(r->name); free(r->seq); r->name = r->seq = NULL; }

static int rd_getc(tho_reader_t *r)
{
    if (r->beg >= r->end) {
        if (r->eof) return -1;
        r->beg = 0; r->end = gzread(r->fp, r->buf, 1 << 16);
        if (r->end <= 0) { r->eof = 1; r->end = 0; return -1; }
    }
    return r->buf[r->beg++];
}

typedef struct { char *s; size_t l, m; } str_t;
static void str_putc(str_t *s, int c)
{
    if (s->l + 2 > s->m) { s->m = s->m ? s->m * 2 : 256; s->s = (char *)realloc(s->s, s->m); }
    s->s[s->l++] = (char)c; s->s[s->l] = 0;
}

int tho_reader_next(tho_reader_t *r, tho_read_t *out)
{
    int c;
    if (r->last == 0) {
        while ((c = rd_getc(r)) != -1 && c != '>' && c != '@') {}
        if (c == -1) return 0;
        r->last = c;
    }
    str_t name = { 0, 0, 0 }, seq = { 0, 0, 0 };
    str_putc(&name, 0); name.l = 0;
    while ((c = rd_getc(r)) != -1 && !isspace(c)) str_putc(&name, c);
    if (c != '\n' && c != -1) while ((c = rd_getc(r)) != -1 && c != '\n') {}
    str_putc(&seq, 0); seq.l = 0;
    while ((c = rd_getc(r)) != -1 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        str_putc(&seq, c);
        int d;
        while ((d = rd_getc(r)) != -1 && d != '\n') str_putc(&seq, d);
        if (seq.l > 1 && seq.s[seq.l - 1] == '\r') seq.s[--seq.l] = 0;   /* KS_SEP_LINE strips one trailing CR */
    }
    r->last = (c == '>' || c == '@') ? c : 0;
    if (c == '+') {
        while ((c = rd_getc(r)) != -1 && c != '\n') {}
        size_t ql = 0;
        while (ql < seq.l && (c = rd_getc(r)) != -1) if (c != '\n') ++ql;
        r->last = 0;
    }
    out->name = name.s; out->seq = seq.s; out->len = (int)seq.l;
    return 1;
}

static int load_one(const char *fn, char **seq)
{
    tho_reader_t *r = tho_reader_open(fn);
    if (!r) { fprintf(stderr, "[tho] fail to open %s\n", fn); exit(1); }
    tho_read_t rd;
    int len = 0;
    if (tho_reader_next(r, &rd) && rd.len > 0) { *seq = rd.seq; len = rd.len; free(rd.name); }
    else fprintf(stderr, "[get_seq_from_fx] No sequence found.\n");
    tho_reader_close(r);
    return len;
}

int tho_para_load_adapters(tho_para_t *p, const char *five_fn, const char *three_fn)
{
    if (!five_fn || !three_fn) return 0;       /* src/main.c:389: both or nothing */
    p->five_len = load_one(five_fn, &p->five_seq);
    p->five_rc_seq = tho_rc_seq(p->five_seq, p->five_len);
    p->three_len = load_one(three_fn, &p->three_seq);
    p->three_rc_seq = tho_rc_seq(p->three_seq, p->three_len);
    return 1;
}
void tho_para_free_adapters(tho_para_t *p)
{
    free(p->five_seq); free(p->five_rc_seq); free(p->three_seq); free(p->three_rc_seq);
    p->five_seq = p->five_rc_seq = p->three_seq = p->three_rc_seq = NULL;
}

/* ---- output, src/main.c:214-271 ---- */
static void put_subpos(FILE *fp, const tho_cons_t *c, char term)
{
    int i;
    fprintf(fp, "%d", c->sub_pos[0] + 2);
    for (i = 1; i < c->pos_n - 1; ++i) fprintf(fp, ",%d", c->sub_pos[i] + 2);
    fprintf(fp, ",%d%c", c->sub_pos[i] + 1, term);
}

void tho_output(FILE *fp, const tho_read_t *rd, const tho_read_res_t *res, const tho_para_t *p)
{
    for (int ci = 0; ci < res->n; ++ci) {
        const tho_cons_t *c = &res->a[ci];
        if (p->only_unit) {
            for (int i = 0; i < c->pos_n - 1; ++i) {
                if (p->out_fmt == THO_FASTA_FMT) {
                    fprintf(fp, ">%s_rep%d_sub%d\n", rd->name, ci, i);
                    for (int j = c->sub_pos[i] + 1; j <= c->sub_pos[i + 1]; ++j) fputc(rd->seq[j], fp);
                    fputc('\n', fp);
                } else if (p->out_fmt == THO_TAB_FMT) {
                    fprintf(fp, "%s\trep%d\tsub%d\t", rd->name, ci, i);
                    for (int j = c->sub_pos[i] + 1; j < c->sub_pos[i + 1]; ++j) fputc(rd->seq[j], fp); /* '<': reference quirk, :224 */
                    fputc('\n', fp);
                }
            }
            continue;
        }
        if (p->out_fmt == THO_FASTA_FMT || p->out_fmt == THO_FASTQ_FMT) {
            fprintf(fp, "%c%s_rep%d_%.1f %d_%d_%d_%d_%.1f_%d_", p->out_fmt == THO_FASTA_FMT ? '>' : '@', rd->name, ci, c->copy_num,
                    rd->len, c->start + 1, c->end + 1, c->cons_len, c->ave_match, c->full_length);
            put_subpos(fp, c, '\n');
        } else {
            fprintf(fp, "%s\trep%d\t%.1f\t%d\t%d\t%d\t%d\t%.1f\t%d\t", rd->name, ci, c->copy_num,
                    rd->len, c->start + 1, c->end + 1, c->cons_len, c->ave_match, c->full_length);
            put_subpos(fp, c, '\t');
        }
        fwrite(c->seq, 1, c->cons_len, fp);
        if (p->out_fmt == THO_FASTQ_FMT) fputs("\n+\n", fp);
        else if (p->out_fmt == THO_TAB_QUAL_FMT) fputc('\t', fp);
        if ((p->out_fmt == THO_FASTQ_FMT || p->out_fmt == THO_TAB_QUAL_FMT) && c->qual) fwrite(c->qual, 1, c->cons_len, fp);
        fputc('\n', fp);
    }
}

/* ---- whole-file driver: chunks of reads, one ticket per read (src/main.c:273-291,402-425) ----
 * Same work-queue design as the reference (a shared ticket counter, one read per ticket), made to scale to the
 * hundreds of cores of a GPU host: the workers are created once and meet the main thread at two barriers per chunk
 * instead of being created and joined per chunk, the ticket is an atomic increment instead of a lock, and the big
 * per-alignment buffers are per-thread (tho_poa.c, tho_ksw2.c). */
#define CHUNK 16384
typedef struct { const tho_para_t *p; tho_read_t *rd; tho_read_res_t *res; volatile int n; int next; volatile int quit; pthread_barrier_t start, done; } work_t;

static void work_chunk(work_t *w)
{
    for (;;) {
        int i = __atomic_fetch_add(&w->next, 1, __ATOMIC_RELAXED);
        if (i >= w->n) break;
        tho_core(w->rd[i].seq, w->rd[i].len, w->p, &w->res[i]);
    }
}
static void *worker(void *a)
{
    work_t *w = (work_t *)a;
    for (;;) {
        pthread_barrier_wait(&w->start);
        if (w->quit) break;
        work_chunk(w);
        pthread_barrier_wait(&w->done);
    }
    return NULL;
}

int tho_run_file(const char *fn, const tho_para_t *p, FILE *out)
{
    tho_reader_t *r = tho_reader_open(fn);
    if (!r) { fprintf(stderr, "[tho] fail to open %s\n", fn); return -1; }
    tho_read_t *rd = (tho_read_t *)calloc(CHUNK, sizeof(tho_read_t));
    tho_read_res_t *res = (tho_read_res_t *)calloc(CHUNK, sizeof(tho_read_res_t));
    int nt = p->n_thread < 1 ? 1 : p->n_thread, total = 0;
    work_t w; memset(&w, 0, sizeof(w));
    w.p = p; w.rd = rd; w.res = res;
    pthread_t *tid = NULL;
    if (nt > 1) {
        pthread_barrier_init(&w.start, NULL, (unsigned)nt + 1); pthread_barrier_init(&w.done, NULL, (unsigned)nt + 1);
        tid = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
        for (int t = 0; t < nt; ++t) pthread_create(&tid[t], NULL, worker, &w);
    }
    for (;;) {
        int n = 0;
        while (n < CHUNK && tho_reader_next(r, &rd[n])) ++n;
        if (n == 0) break;
        w.n = n; w.next = 0;
        if (nt == 1) work_chunk(&w);
        else { pthread_barrier_wait(&w.start); pthread_barrier_wait(&w.done); }
        for (int i = 0; i < n; ++i) {
            if (out) tho_output(out, &rd[i], &res[i], p);
            tho_read_res_clear(&res[i]); tho_read_free(&rd[i]);
        }
        total += n;
    }
    if (nt > 1) {
        w.quit = 1; pthread_barrier_wait(&w.start);
        for (int t = 0; t < nt; ++t) pthread_join(tid[t], NULL);
        pthread_barrier_destroy(&w.start); pthread_barrier_destroy(&w.done); free(tid);
    }
    for (int i = 0; i < CHUNK; ++i) tho_read_res_free(&res[i]);
    free(rd); free(res);
    tho_reader_close(r);
    return total;
}
