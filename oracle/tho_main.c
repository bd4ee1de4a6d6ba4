/*
 * oracle/tho_main.c -- TEST INFRASTRUCTURE (see tho.h).
 * Command line of the CPU oracle; option surface restates src/main.c:16-52,438-535
 * (including its quirks: --gap_open/--gap_ext/--ada-match-rat long names, -s without a long
 * form, K/M/G suffixes only on -p/-P, -r < 1 is a fraction).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <getopt.h>
#include <sys/time.h>
#include <sys/resource.h>
#include <malloc.h>
#include "tho.h"

static const struct option lopt[] = {
    { "kmer-length", 1, NULL, 'k' }, { "window-size", 1, NULL, 'w' }, { "HPC-kmer", 0, NULL, 'H' },
    { "min-copy", 1, NULL, 'c' }, { "max-diverg", 1, NULL, 'e' }, { "min-period", 1, NULL, 'p' }, { "max-period", 1, NULL, 'P' },
    { "match", 1, NULL, 'M' }, { "mismatch", 1, NULL, 'X' }, { "gap_open", 1, NULL, 'O' }, { "gap_ext", 1, NULL, 'E' },
    { "five-prime", 1, NULL, '5' }, { "three-prime", 1, NULL, '3' }, { "ada-match-rat", 1, NULL, 'a' },
    { "output", 1, NULL, 'o' }, { "min-len", 1, NULL, 'm' }, { "min-cov", 1, NULL, 'r' }, { "unit-seq", 0, NULL, 'u' },
    { "longest", 0, NULL, 'l' }, { "full-len", 0, NULL, 'F' }, { "out-fmt", 1, NULL, 'f' }, { "thread", 1, NULL, 't' },
    { "help", 0, NULL, 'h' }, { "version", 0, NULL, 'v' }, { 0, 0, 0, 0 }
};

static int64_t parse_num(const char *s)
{
    char *e; double x = strtod(s, &e);
    if (*e == 'G' || *e == 'g') x *= 1e9; else if (*e == 'M' || *e == 'm') x *= 1e6; else if (*e == 'K' || *e == 'k') x *= 1e3;
    return (int64_t)(x + .499);
}

static double now(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + t.tv_usec * 1e-6; }

int main(int argc, char **argv)
{
    tho_para_t p; tho_para_init(&p);
    const char *five = NULL, *three = NULL; FILE *out = stdout; char *s; double x, t0 = now();
    int c, quiet_out = 0;
    while ((c = getopt_long(argc, argv, "k:w:m:Hhvc:e:p:P:M:X:E:O:5:3:a:o:ur:qslFf:t:", lopt, NULL)) >= 0) {
        switch (c) {
        case 'h': fprintf(stderr, "tho_oracle: CPU restatement of TideHunter (test infrastructure); options as TideHunter 1.5.5\n"); return 1;
        case 'v': printf("1.5.5-oracle\n"); return 0;
        case 'k': p.k = atoi(optarg); if (p.k > 16) { fprintf(stderr, "\n[main] Error: k-mer length can not be larger than 16 (%d).\n", p.k); return 0; } break;
        case 'w': p.w = atoi(optarg); break;
        case 'H': p.hpc = 1; break;
        case 'c': p.min_copy = atoi(optarg); if (p.min_copy < 2) { fprintf(stderr, "Error: -c --min-copy needs to be >= 2. (%d)\n", p.min_copy); return 0; } break;
        case 'e': p.max_div = atof(optarg); break;
        case 'p': p.min_p = parse_num(optarg); if (p.min_p < 2) { fprintf(stderr, "Error: -p --min-period needs to be >= 2.\n"); return 0; } break;
        case 'P': p.max_p = parse_num(optarg); if (p.max_p > 4294967295LL) { fprintf(stderr, "Error: -P --max-period needs to be <= 4294967295.\n"); return 0; } break;
        case 'M': p.match = atoi(optarg); break;
        case 'X': p.mismatch = atoi(optarg); break;
        case 'O': p.gap_open1 = strtol(optarg, &s, 10); if (*s == ',') p.gap_open2 = strtol(s + 1, &s, 10); break;
        case 'E': p.gap_ext1 = strtol(optarg, &s, 10); if (*s == ',') p.gap_ext2 = strtol(s + 1, &s, 10); break;
        case '5': five = optarg; break;
        case '3': three = optarg; break;
        case 'a': p.ada_match_rat = atof(optarg); break;
        case 'o': if (!strcmp(optarg, "/dev/null")) quiet_out = 1; out = fopen(optarg, "w"); if (!out) { perror(optarg); return 1; } break;
        case 'm': p.min_len = atoi(optarg); break;
        case 'r': x = strtod(optarg, &s); if (x < 1.0) { p.min_frac = x; p.min_cov = 0; } else { p.min_cov = (int)(x + .499); p.min_frac = 0.0; } break;
        case 'u': p.only_unit = 1; break;
        case 'l': p.only_longest = 1; break;
        case 'F': p.only_full_length = 1; break;
        case 's': p.single_copy = 1; break;
        case 'f': p.out_fmt = atoi(optarg); if (p.out_fmt < 1 || p.out_fmt > 4) { fprintf(stderr, "\n[main] Error: unknown format number. (-%c)\n", c); return 0; } break;
        case 't': p.n_thread = atoi(optarg); break;
        default: return 0;   /* includes 'q' (in the optstring, no case) */
        }
    }
    (void)quiet_out;
    if (p.only_unit && (p.out_fmt == THO_FASTQ_FMT || p.out_fmt == THO_TAB_QUAL_FMT)) { fprintf(stderr, "\n[main] Error: unit sequences can only be ouput in FASTA or TAB format.\n"); return 0; }
    if (optind + 1 > argc) { fprintf(stderr, "\n[main] Error: please specify an input file.\n"); return 0; }
    if (p.only_full_length && (!five || !three)) { fprintf(stderr, "\n[main] Error: 5' and 3' adapter sequence need to be provided.\n"); return 0; }
    if (p.w < 1 || p.w > 256) { fprintf(stderr, "[tho_oracle] -w must be in 1..256 (the reference's window array has 256 slots)\n"); return 2; }
    tho_para_load_adapters(&p, five, three);
    /* the pin kit (tests/test_oracle_ref_full.py) runs the oracle with either consensus tie rule without a rebuild */
    if (getenv("THO_POA_GAP_TIE")) tho_poa_set_opt(THO_POA_OPT_GAP_TIE, atoi(getenv("THO_POA_GAP_TIE")));
    /* many worker threads: keep freed memory in the per-thread malloc arenas instead of returning it to the kernel (every
     * mmap/munmap/trim takes the process-wide mm lock, which is what limited the scaling of this program on a 256-core host) */
    mallopt(M_MMAP_THRESHOLD, 32 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 16 << 20);
    const double t_run = now();
    int n = tho_run_file(argv[optind], &p, out);
    const double wall = now() - t_run;
    struct rusage ru; getrusage(RUSAGE_SELF, &ru);
    const double cpu = ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec);
    fprintf(stderr, "[tho_oracle] reads: %d; Real time: %.3f sec; CPU: %.3f sec; Peak RSS: %.3f GB\n", n, now() - t0, cpu, ru.ru_maxrss / 1024.0 / 1024.0);
    if (n > 0) fprintf(stderr, "[tho_oracle] threads: %d; %.1f reads/s; %.2f ms per read per core (CPU time / reads); %.2f ms per read x threads (wall x threads / reads)\n",
                       p.n_thread < 1 ? 1 : p.n_thread, n / wall, 1000.0 * cpu / n, 1000.0 * wall * (p.n_thread < 1 ? 1 : p.n_thread) / n);
    tho_para_free_adapters(&p);
    if (out != stdout) fclose(out);
    return 0;
}
