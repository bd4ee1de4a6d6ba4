/* tests/c_caller/caller.c -- a plain C program against include/tidehunter_amd.h, no Python anywhere:
 *   th_init_para -> th_init -> th_alloc_tandem_seq -> th_process_batch -> th_output -> th_destroy
 * i.e. the body of mini_tandem() (src/main.c:375-436) with the pthread fan-out replaced by the batch seam.
 * Reads FASTA from argv[1].  Exit status 3 = th_init failed (no GPU: there is no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include "tidehunter_amd.h"

/* the layouts the reference's own structs have (src/tidehunter.h:47-72 on LP64): a maintainer may cast */
_Static_assert(offsetof(th_para_t, hpc) == 16 && offsetof(th_para_t, min_p) == 56 && offsetof(th_para_t, match) == 80, "mini_tandem_para layout");
_Static_assert(offsetof(th_para_t, five_fn) == 112 && offsetof(th_para_t, three_fn) == 144 && offsetof(th_para_t, out_fmt) == 172, "mini_tandem_para layout");
_Static_assert(offsetof(th_para_t, cons_out) == 200 && offsetof(th_para_t, n_thread) == 216 && sizeof(th_para_t) == 224, "mini_tandem_para layout");
_Static_assert(sizeof(th_seq_t) == 96 && offsetof(th_seq_t, qual) == 72, "seq_t layout");
_Static_assert(offsetof(th_tandem_seq_t, cons_n) == 8 && offsetof(th_tandem_seq_t, cons_start) == 16 && offsetof(th_tandem_seq_t, sub_pos) == 64 && sizeof(th_tandem_seq_t) == 96, "tandem_seq_t layout");

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: caller in.fa\n"); return 2; }
    FILE *fp = fopen(argv[1], "r");
    if (!fp) { perror(argv[1]); return 2; }
    enum { MAXR = 64 };
    static char *names[MAXR], *seqs[MAXR]; static int lens[MAXR];
    int n = 0; char *line = NULL; size_t cap = 0; ssize_t got;
    while ((got = getline(&line, &cap, fp)) > 0 && n < MAXR) {
        while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = 0;
        if (line[0] == '>') {
            if (names[n] && seqs[n]) ++n;
            if (n == MAXR) break;
            names[n] = strdup(line + 1); char *sp = strchr(names[n], ' '); if (sp) *sp = 0;
        } else if (got > 0 && names[n]) {          /* sequence lines of one record are concatenated */
            seqs[n] = (char *)realloc(seqs[n], (size_t)lens[n] + (size_t)got + 1);
            memcpy(seqs[n] + lens[n], line, (size_t)got + 1); lens[n] += (int)got;
        }
    }
    if (n < MAXR && names[n] && seqs[n]) ++n;
    fclose(fp); free(line);

    th_para_t *p = th_init_para();                     /* mini_tandem_init_para() */
    p->out_fmt = TH_TAB_FMT;
    th_handle_t *h = th_init(p, 0);                    /* aux_init() */
    if (!h) { fprintf(stderr, "caller: th_init failed: %s\n", th_last_error()); th_free_para(p); return 3; }
    th_tandem_seq_t *ts = th_alloc_tandem_seq(n);      /* alloc_tandem_seq() */
    if (th_process_batch(h, (const char *const *)seqs, lens, n, ts) != 0) { fprintf(stderr, "caller: %s\n", th_last_error()); return 4; }
    th_output(stdout, n, (const char *const *)names, (const char *const *)seqs, lens, ts, p);   /* mini_tandem_output() */
    th_free_tandem_seq(ts, n); th_destroy(h); th_free_para(p);
    for (int i = 0; i < n; ++i) { free(names[i]); free(seqs[i]); }
    return 0;
}
