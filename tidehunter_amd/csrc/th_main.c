/*
 * th_main.c -- command line of the MI355X TideHunter build ("TideHunter-amd").
 * Option surface restates main(), src/main.c:438-535 and the table at :16-52, including its
 * quirks (--gap_open / --gap_ext / --ada-match-rat long names, -s has no long form, K/M/G suffixes
 * only on -p/-P, -r < 1 is a fraction, 'q' sits in the optstring without a case and ends the run).
 * -t is accepted and ignored (the batch goes to the GPU); -g / TH_DEVICE picks the device.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <getopt.h>
#include <math.h>
#include <sys/time.h>
#include <sys/resource.h>
#include "tidehunter_amd.h"

static const struct option lopt[] = {
    { "kmer-length", 1, NULL, 'k' }, { "window-size", 1, NULL, 'w' }, { "HPC-kmer", 0, NULL, 'H' },
    { "min-copy", 1, NULL, 'c' }, { "max-diverg", 1, NULL, 'e' }, { "min-period", 1, NULL, 'p' }, { "max-period", 1, NULL, 'P' },
    { "match", 1, NULL, 'M' }, { "mismatch", 1, NULL, 'X' }, { "gap_open", 1, NULL, 'O' }, { "gap_ext", 1, NULL, 'E' },
    { "five-prime", 1, NULL, '5' }, { "three-prime", 1, NULL, '3' }, { "ada-match-rat", 1, NULL, 'a' },
    { "output", 1, NULL, 'o' }, { "min-len", 1, NULL, 'm' }, { "min-cov", 1, NULL, 'r' }, { "unit-seq", 0, NULL, 'u' },
    { "longest", 0, NULL, 'l' }, { "full-len", 0, NULL, 'F' }, { "out-fmt", 1, NULL, 'f' }, { "thread", 1, NULL, 't' },
    { "gpu", 1, NULL, 'g' }, { "batch", 1, NULL, 'b' },
    { "help", 0, NULL, 'h' }, { "version", 0, NULL, 'v' }, { 0, 0, 0, 0 }
};

static int64_t parse_num(const char *s)   /* th_parse_num, src/main.c:54-63 */
{
    char *e; double x = strtod(s, &e);
    if (*e == 'G' || *e == 'g') x *= 1e9; else if (*e == 'M' || *e == 'm') x *= 1e6; else if (*e == 'K' || *e == 'k') x *= 1e3;
    return (int64_t)(x + .499);
}

static int usage(void)
{
    fprintf(stderr, "\nTideHunter-amd: Tandem repeats detection and consensus calling from noisy long reads (MI355X/HIP hot path)\n\n");
    fprintf(stderr, "Version: %s\n\nUsage:   TideHunter-amd [options] in.fa/fq > cons.fa\n\n", th_version());
    fprintf(stderr, "Options: as TideHunter 1.5.5 (-k -w -H -c -e -p -P -M -X -O -E -5 -3 -a -o -m -r -u -l -F -s -f -t -h -v)\n");
    fprintf(stderr, "         -g --gpu LIST   HIP device ordinal(s), e.g. 0 or 0,1,2,3: the listed devices share the batches [0];  -b --batch INT   reads per device batch [65536]\n");
    fprintf(stderr, "Note:    every stage is bit-identical to this repository's CPU restatement of TideHunter 1.5.5 (oracle/), which reproduces every known answer in the\n"
                    "         reference tree; the ksw2 / edlib results are additionally identical to the reference's compiled code.  Consensus calling implements the\n"
                    "         published abPOA algorithm (libabpoa itself is not linked and could not be compared): consensus bases of noisy units may differ from upstream.\n");
    return 1;
}

static double now(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + t.tv_usec * 1e-6; }

int main(int argc, char **argv)
{
    setenv("TH_BLOCKING_DEVICE", "1", 0);      /* this process is ours: blocking waits in the runtime's slow paths too (the library never sets the device flag by itself) */
    th_para_t *p = th_init_para();
    FILE *out = stdout; char *s; double x, t0 = now();
    int c, batch = 0, rc = 0;
    int devices[16] = { getenv("TH_DEVICE") ? atoi(getenv("TH_DEVICE")) : 0 }, n_dev = 1;
    while ((c = getopt_long(argc, argv, "k:w:m:Hhvc:e:p:P:M:X:E:O:5:3:a:o:ur:qslFf:t:g:b:", lopt, NULL)) >= 0) {
        switch (c) {
        case 'h': return usage();
        case 'v': printf("%s\n", th_version()); goto END;
        case 'k': p->k = atoi(optarg); if (p->k > 16) { fprintf(stderr, "\n[main] Error: k-mer length can not be larger than 16 (%ld).\n", (long)p->k); goto END; } break;
        case 'w': p->w = atoi(optarg); break;
        case 'H': p->hpc = 1; break;
        case 'c': p->min_copy = atoi(optarg); if (p->min_copy < 2) { fprintf(stderr, "Error: -c --min-copy needs to be >= 2. (%d)\n", p->min_copy); goto END; } break;
        case 'e': p->max_div = atof(optarg); break;
        case 'p': p->min_p = parse_num(optarg); if (p->min_p < 2) { fprintf(stderr, "Error: -p --min-period needs to be >= 2. (%ld)\n", (long)p->min_p); goto END; } break;
        case 'P': p->max_p = parse_num(optarg); if (p->max_p > 4294967295LL) { fprintf(stderr, "Error: -P --max-period needs to be <= 4294967295. (%ld)\n", (long)p->max_p); goto END; } break;
        case 'M': p->match = atoi(optarg); break;
        case 'X': p->mismatch = atoi(optarg); break;
        case 'O': p->gap_open1 = strtol(optarg, &s, 10); if (*s == ',') p->gap_open2 = strtol(s + 1, &s, 10); break;
        case 'E': p->gap_ext1 = strtol(optarg, &s, 10); if (*s == ',') p->gap_ext2 = strtol(s + 1, &s, 10); break;
        case '5': p->five_fn = strdup(optarg); break;
        case '3': p->three_fn = strdup(optarg); break;
        case 'a': p->ada_match_rat = atof(optarg); break;
        case 'o': out = fopen(optarg, "w"); if (!out) { perror(optarg); rc = 1; goto END; } break;
        case 'm': p->min_len = atoi(optarg); break;
        case 'r': x = strtod(optarg, &s); if (x < 1.0) { p->min_frac = x; p->min_cov = 0; } else { p->min_cov = (int)(x + .499); p->min_frac = 0.0; } break;
        case 'u': p->only_unit = 1; break;
        case 'l': p->only_longest = 1; break;
        case 'F': p->only_full_length = 1; break;
        case 's': p->single_copy = 1; break;
        case 'f': p->out_fmt = atoi(optarg); if (p->out_fmt < 1 || p->out_fmt > 4) { fprintf(stderr, "\n[main] Error: unknown format number. (-%c)\n", c); goto END; } break;
        case 't': p->n_thread = atoi(optarg); break;
        case 'g': n_dev = 0; for (char *tok = strtok(optarg, ","); tok && n_dev < 16; tok = strtok(NULL, ",")) devices[n_dev++] = atoi(tok); if (n_dev == 0) { devices[0] = 0; n_dev = 1; } break;
        case 'b': batch = atoi(optarg); break;
        default: goto END;
        }
    }
    if (p->only_unit && (p->out_fmt == TH_FASTQ_FMT || p->out_fmt == TH_TAB_QUAL_FMT)) { fprintf(stderr, "\n[main] Error: unit sequences can only be ouput in FASTA or TAB format.\n"); usage(); goto END; }
    if (optind + 1 > argc) { fprintf(stderr, "\n[main] Error: please specify an input file.\n"); usage(); goto END; }
    if (p->only_full_length && (p->five_fn == NULL || p->three_fn == NULL)) { fprintf(stderr, "\n[main] Error: 5' and 3' adapter sequence need to be provided.\n"); usage(); goto END; }
    if (p->five_fn == NULL && p->three_fn != NULL) fprintf(stderr, "\n[main] Warning: only 3' adapter sequence is provided. Full-length sequence cannot be determined.\n");
    if (p->five_fn != NULL && p->three_fn == NULL) fprintf(stderr, "\n[main] Warning: only 5' adapter sequence is provided. Full-length sequence cannot be determined.\n");
    p->div_exp = exp(2 * p->k * p->max_div);
    {
        int n = th_run_file_multi(argv[optind], p, out, devices, n_dev, batch);
        if (n < 0) { fprintf(stderr, "[main] Error: %s\n", th_last_error()); rc = 1; goto END; }
        struct rusage ru; getrusage(RUSAGE_SELF, &ru);
        fprintf(stderr, "[main] Real time: %.3f sec; CPU: %.3f sec; Peak RSS: %.3f GB\n", now() - t0,
                ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec), ru.ru_maxrss / 1024.0 / 1024.0);
    }
END:
    if (out && out != stdout) fclose(out);
    th_free_para(p);
    return rc;
}
