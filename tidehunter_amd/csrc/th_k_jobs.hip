// th_k_jobs.hip -- the consensus job tables, built on the device (gfx950).
//
// What seqs_msa() does with the partition of one chain (src/gen_cons.c:192-205: split par_pos at the -1 separators, keep runs
// with more than min_copy boundaries) and abpoa_gen_cons() with the units of one run (src/abpoa_cons.c:36-48: slice the units,
// drop those that touch the end of the read), as three small kernels per batch instead of a host loop in the middle of the
// pipeline (the reference has no hop between its stages either, src/main.c:402-425):
//   th_k_jobs_count   one thread per read: how many par_pos entries, jobs, units, consensus bytes and alignment slots its chains need
//   th_k_scan_rows    one workgroup: exclusive prefix sums of those counters over the reads, totals into the counter block
//   th_k_jobs_emit    one thread per read: writes its chains' par_pos compactly, its ThTask / ThUnit / th_job_t entries and the
//                     task lists of the two POA kernel variants at the offsets the scan assigned -- so the tables come out in
//                     read order (chains in emission order, runs left to right), exactly as the host loop produced them
// and, after the POA, the same pattern per JOB for the alignment item lists and the compacted consensus offsets
// (th_k_items_count / th_k_scan_rows / th_k_items_emit).  The host only reads the 256-byte counter block.
#include "th_dev.h"

#define TH_JOB_COLS 7             // per-read counters: par_pos entries, jobs, units, consensus bytes, alignment slots, jobs of the int16 / of the relative-int16 POA variant
#define TH_ITEM_COLS 4            // per-job counters: small / mid / big alignment items, compacted consensus bytes

// Scratch of one POA wave for a job, in ints.  Mirrors the carve in th_k_poa: 39 node / row arrays of `cap`, the overflow pool,
// the alignment-op arrays, then the cell region.  Cell region: every band cell has a 2-byte direction word; rows that store full
// cells add 16 bytes per cell -- a few per cent of the rows when the band fits the LDS ring, EVERY row when it does not (the 32-bit
// variant on long units: measured band ~ 2w + 1 + len / 30 at 15 % error).  Graph rows grow by ~9 % of the unit length per unit
// (measured: 1.27 x at 4 units of 5000, ~1.5 x at 9 units of 1000, ~2.6 x at 20 units of 200).  cell_factor is the attempt's
// safety factor (1.5 at first, doubled for the jobs that come back with "scratch too small").
#ifndef TH_POA_CELL_INTS
#define TH_POA_CELL_INTS (TH_POA_D8 ? 1.25 : 1.5)     // ints reserved per estimated band cell (x the attempt's factor); 1.0 sent 329 of 100 000 headline jobs to the rerun
#endif
__host__ __device__ static inline long long th_poa_need_ints(int n_units, int sum_len, int max_len, double cell_factor, bool noband, bool wide /* class 0: rows wider than the ring store full cells */)
{
    if (n_units <= 2) return 0;
    int fexp = 0; for (double f = 1.5; f * 1.99 < cell_factor; f *= 2) ++fexp;        // the attempt's factor as the kernel is told it: 1.5 * 2^fexp
    const ThPoaCaps K = th_poa_caps(n_units, sum_len, max_len, fexp, noband);
    const long long cap = K.cap, ocap = K.ocap, opn = K.opn;
    const long long graph = 40 * cap + 2 * ocap + 5 * opn + ((long long)sum_len + 16) + 512;      // (round 5: + the units' node paths and the nodes' consensus columns, th_poa_carve)
    if (noband) return (graph + (long long)(cap * ((long long)max_len + 4) * (cell_factor > 1.5 ? 1.25 : 1.0) * 4.5) + 63) & ~63LL;   // full rows, all stored in full
    const long long w = TH_POA_WB + max_len / 100;
    const long long span = 2 * w + 1 + 48 + max_len / 30;
    long long rows = (long long)(max_len * (1.0 + 0.09 * (n_units - 1))) + 64;
    if (rows > sum_len + 2) rows = sum_len + 2;
    const bool all_full = wide && span + 16 >= TH_POA_RW_N;
    // ints per band cell: a direction word of two bytes (one for three rows in four with TH_POA_D8) plus the full cells of the few rows a
    // far row will read -- ~3.3 bytes on average (2.3 with TH_POA_D8); TH_POA_CELL_INTS x 4 bytes (x the attempt's factor) are reserved
    return (graph + (long long)(rows * span * cell_factor * (all_full ? 4.5 : TH_POA_CELL_INTS)) + 63) & ~63LL;
}
// the factor exponent of a POA launch's flags (bits 1..4) from the attempt's factor
static inline int th_poa_fexp(double cell_factor) { int fexp = 0; for (double f = 1.5; f * 1.99 < cell_factor; f *= 2) ++fexp; return fexp > 15 ? 15 : fexp; }
// Which POA variant a job starts on.  The int16 variant gives a job back (status 7) when an alignment's scores may not fit
// (th_poa_dp: cache16_ok, evaluated with the graph's row count at that alignment); with the row estimate above the same test
// tells beforehand, for nearly every job, that it would -- those start on the 32-bit variant (both are exact: same results).
// Returns the job's class: 1 narrow (thpoa_n int16, absolute scores), 2 long (thpoa_l: relative int16 scores, rows of up to 384 columns,
// units of up to TH_POA_QCAP_LONG bases), 0 wide (32-bit rows).
__host__ __device__ static inline int th_poa_job_class(int n_units, int max_len, const ThDevPara &P)
{
    const int M = P.match, X = P.mismatch, maxE = P.ge1 > P.ge2 ? P.ge1 : P.ge2;
    const int g1 = P.go1 + P.ge1, g2 = P.go2 + P.ge2, gap1 = g1 < g2 ? g1 : g2;
    const int mx = M > X ? (M > gap1 ? M : gap1) : (X > gap1 ? X : gap1);
    if (M > 31 || X > 31) return 0;
    if (max_len <= TH_POA_QCAP) {
        const long long rows = (long long)(max_len * (1.0 + 0.09 * (n_units - 1))) + 64;
        const long long real_b = (long long)mx * (max_len + rows) + P.go1 + P.go2 + 2 * maxE;
        if (real_b + (long long)M * rows + 2LL * (TH_POA_RW_N + 64) * maxE < 32000) return 1;
    }
    // (the range test of th_poa_dp's relative mode: it depends on the scoring only)
    if (max_len <= TH_POA_QCAP_LONG &&
        (long long)mx * 3 * (TH_POA_RW_L + 64) + 2LL * (TH_POA_RW_L + 64) * maxE + P.go1 + P.go2 < -TH_POA_REL_THR - TH_POA_REL_HI) return 2;
    return 0;
}

struct ThJobTables {
    const uint32_t *roff; const int *read_ch_base, *read_ch_n;
    const int *par, *par_off, *par_n;                // the partition's arena and each chain's region in it
    int *par_dst, *parc;                             // compact copy: chain c's boundaries at parc[par_dst[c] ..]
    ThTask *tasks; ThUnit *units; th_job_t *jobs; int *list_n, *list_w, *list_l;
    int task_cap, unit_cap, parc_cap, cls_mask;
    int fexp0[3];                                    // per POA class (0 wide, 1 narrow, 2 long): the first attempt's scratch factor is 1.5 * 2^fexp0 (th_pipeline.hip: poa_fexp0)
};

// the walk over one read's chains; EMIT = false counts into c[], EMIT = true writes at the offsets in c[] (advancing them)
template <bool EMIT>
__device__ __forceinline__ void th_jobs_walk(const ThJobTables &J, const ThDevPara &P, int r, int only_unit, int *c, ThCnt *C)
{
    const int cls_mask = J.cls_mask;
    const int L = (int)(J.roff[r + 1] - J.roff[r]);
    const int min_copy = P.min_copy;
    const int cb = J.read_ch_base[r], cn = J.read_ch_n[r];
    for (int ch = cb; ch < cb + cn; ++ch) {
        const int par_n = J.par_n[ch]; const int *pp = J.par + J.par_off[ch];
        const int dst = c[0];
        if (EMIT) { J.par_dst[ch] = dst; for (int i = 0; i < par_n; ++i) J.parc[dst + i] = pp[i]; }
        c[0] += par_n;
        if (par_n < min_copy + 1) continue;                            // src/tidehunter.c:42
        int i = 0;
        while (i < par_n - min_copy) {
            if (pp[i] < 0) { ++i; continue; }
            int j; for (j = i + 1; j < par_n; ++j) if (pp[j] < 0) break;
            if (j - i > min_copy) {
                int n_units = 0, sum_len = 0, max_len = 0, first_len = 0;
                for (int u = i; u < j - 1; ++u) {                      // src/abpoa_cons.c:41-48
                    const int start = pp[u], end = pp[u + 1];
                    if (start < 0 || end < 0 || start >= L - 1 || end + 1 > L) continue;
                    const int len = end - start;
                    if (EMIT) { ThUnit x = { start + 1, len, u - i, 0 }; J.units[c[2] + n_units] = x; }
                    if (n_units == 0) first_len = len;
                    ++n_units; sum_len += len; max_len = max(max_len, len);
                }
                const int cons_cap = n_units <= 2 ? first_len : min(sum_len, 3 * max_len + 64);
                int cls = th_poa_job_class(n_units, max_len, P) & cls_mask;       // (cls_mask 1: no long class -- TH_POA_LONG=0)
                if ((J.cls_mask & 4) && cls == 1) cls = 2;                        // (TH_POA_FORCE_LONG=1, tests: short units on the long-unit variant)
                if (EMIT) {
                    const int t = c[1];
                    ThTask T; T.read = r; T.pos_off = dst + i; T.pos_n = j - i; T.unit_off = c[2]; T.n_units = n_units; T.sum_len = sum_len; T.max_len = max_len;
                    T.cons_off = c[3]; T.cons_cap = cons_cap; T.aln_off = c[4]; T.narrow = cls; T.noband = 0;
                    J.tasks[t] = T;
                    th_job_t jb; jb.read = r; jb.pos_off = T.pos_off; jb.pos_n = T.pos_n; jb.n_seqs = n_units; jb.cons_off = 0; jb.cons_len = 0; jb.aln_off = T.aln_off; jb.status = 0;
                    J.jobs[t] = jb;
                    if (cls == 1) J.list_n[c[5]] = t; else if (cls == 2) J.list_l[c[6]] = t; else J.list_w[t - c[5] - c[6]] = t;
                    if (!only_unit) {
                        if (n_units <= 1) atomicOr(&C->err, TH_E_MSA_UNITS);
                        atomicMax(cls == 1 ? &C->need_n : cls == 2 ? &C->need_l : &C->need_w, (unsigned long long)th_poa_need_ints(n_units, sum_len, max_len, 1.5 * (double)(1 << J.fexp0[cls]), false, cls == 0));
                        atomicMax(&C->max_unit, max_len); atomicMax(&C->max_cons_cap, cons_cap);
                    }
                }
                c[1] += 1; c[2] += n_units; c[3] += (cons_cap + 15) & ~15; c[4] += j - i - 1; c[5] += cls == 1 ? 1 : 0; c[6] += cls == 2 ? 1 : 0;
            }
            i = j + 1;
        }
    }
}

__global__ void __launch_bounds__(256)
th_k_jobs_count(int n_reads, ThJobTables J, ThDevPara P, int only_unit, int *__restrict__ rows, ThCnt *C)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int c[TH_JOB_COLS] = { 0, 0, 0, 0, 0, 0, 0 };
    // the record arena overflowed: th_k_partition returned at entry, so par_n[] / par_off[] hold whatever the batch before left there
    // (or nothing at all on a handle's first batch) -- zero rows, nothing is walked; the host reruns extraction, partition and this build
    if (!(C->err & TH_E_REC_OVERFLOW)) th_jobs_walk<false>(J, P, r, only_unit, c, C);
#pragma unroll
    for (int k = 0; k < TH_JOB_COLS; ++k) rows[(long long)r * TH_JOB_COLS + k] = c[k];
}

__global__ void __launch_bounds__(256)
th_k_jobs_emit(int n_reads, ThJobTables J, ThDevPara P, int only_unit, const int *__restrict__ rows, ThCnt *C)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    if (C->err & (TH_E_JOB_CAP | TH_E_REC_OVERFLOW)) return;      // the tables are too small, or the partition did not run (see th_k_jobs_count): nothing is written, the host reruns the build
    int c[TH_JOB_COLS];
#pragma unroll
    for (int k = 0; k < TH_JOB_COLS; ++k) c[k] = rows[(long long)r * TH_JOB_COLS + k];
    th_jobs_walk<true>(J, P, r, only_unit, c, C);
}

// Exclusive prefix sums of COLS counters over n rows (row-major, in place), one workgroup of 1024 threads: every thread sums a
// contiguous slice of rows, the slice totals are scanned across the workgroup, then every thread rewrites its slice.  The grand
// totals go to totals[0 .. COLS) and, for the job build, totals beyond the table capacities (caps.on) raise TH_E_JOB_CAP.
// n is a few hundred thousand at most: tens of microseconds.
struct ThScanCaps { int on, v[8]; };
template <int COLS>
__global__ void __launch_bounds__(1024)
th_k_scan_rows(const int *__restrict__ n_ptr, int n_fixed, int *__restrict__ rows, int *__restrict__ totals, ThScanCaps caps, int *__restrict__ err)
{
    __shared__ int s_wave[COLS][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = n_ptr ? *n_ptr : n_fixed;
    const int chunk = (n + 1023) / 1024;
    const long long lo = (long long)tid * chunk, hi = min((long long)n, lo + chunk);
    int s[COLS];
#pragma unroll
    for (int k = 0; k < COLS; ++k) s[k] = 0;
    for (long long i = lo; i < hi; ++i)
#pragma unroll
        for (int k = 0; k < COLS; ++k) s[k] += rows[i * COLS + k];
    int ex[COLS];
#pragma unroll
    for (int k = 0; k < COLS; ++k) {
        const int inc = th_wave_incl_sum(s[k]);
        ex[k] = inc - s[k];
        if (lane == 63) s_wave[k][wave] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < COLS; ++k) {
        int before = 0, total = 0;
        for (int w = 0; w < 16; ++w) { const int v = s_wave[k][w]; if (w < wave) before += v; total += v; }
        ex[k] += before;
        if (tid == 0) { totals[k] = total; if (caps.on && total > caps.v[k]) atomicOr(err, TH_E_JOB_CAP); }
    }
    for (long long i = lo; i < hi; ++i)
#pragma unroll
        for (int k = 0; k < COLS; ++k) { const int v = rows[i * COLS + k]; rows[i * COLS + k] = ex[k]; ex[k] += v; }
}

// ---- after the POA: alignment item lists (three staging-area classes) and compacted consensus offsets, per job ----
struct ThAlnItem { int task, which; };   // which >= 0: unit index in the run; -1 left extension; -2 right extension
#define TH_ALN_SMALL 12288
#define TH_ALN_BIG16 16384
#define TH_ALN_BIG 24576
// Jobs whose unit + consensus fit the 12 KB staging area of the two-register variant of the anti-diagonal engine (periods up to ~2500;
// that variant runs at three waves per SIMD by its registers, which leaves 13 KB of LDS per wave) form the first list, the rest a
// third one for the variant with a 16 / 24 KB area and up to 6 registers per lane (two waves per SIMD).  (Round 2 had a 4 KB and a 7 KB
// list: the 4 KB one dated from a variant with five waves per SIMD.)  The two flank extensions of every job go to the first list.
// Round 4: a job whose first band would want three registers per lane or more (units of several thousand bases) ALSO goes to the first list
// when its sequences fit that variant's staging area with ONE register (up to ~5600 + 5600 bases): there it is probed with one register
// and, where that cannot be certified, finished by the pruned strip engine in place -- at three waves per SIMD.  The six-register variant
// ran those probes at two waves per SIMD, a launch of its own that was 162 of config 4's 966 ms (TH_ALN_PROBE_SMALL=0: round 3's routing).
__device__ __forceinline__ int th_aln_class(const ThTask &T, int cl, int wdiv, int wadd, int wtol)
{
    const int probe_small = wtol >> 16; wtol &= 0xffff;                       // (the flag travels in the upper half of wtol)
    const long long need2 = (long long)T.max_len + cl + 512 * 2 + 16, need1 = (long long)T.max_len + cl + 512 + 16;
    const bool two_regs = 2 * (cl / wdiv + wadd - wtol) + 2 + 64 <= 512;     // th_ad_pick would settle for NP <= 2
    if (need2 <= TH_ALN_SMALL && two_regs) return 0;
    return (probe_small && need1 <= TH_ALN_SMALL) ? (probe_small == 2 ? 1 : 0) : 2;      // (2: a list -- and a launch -- of their own on the same variant)
}
__global__ void __launch_bounds__(256)
th_k_items_count(const int *__restrict__ n_tasks, const ThTask *__restrict__ tasks, int *__restrict__ cons_len, const int *__restrict__ status,
                 int wdiv, int wadd, int wtol, int *__restrict__ rows, ThCnt *C)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *n_tasks) return;
    const int st = status[t];
    if (st == 1 || st == 2 || st == 3 || st == 4 || st == 7) atomicAdd(&C->n_retry, 1);
    else if (st != 0) { atomicAdd(&C->n_failed, 1); cons_len[t] = 0; }
    const int cl = st ? 0 : cons_len[t];
    const ThTask T = tasks[t];
    int c[TH_ITEM_COLS] = { 2, 0, 0, (cl + 3) & ~3 };
    const int cls = th_aln_class(T, cl, wdiv, wadd, wtol);
    c[cls] += T.pos_n - 1;
    if (cls == 2) atomicMax(&C->max_need_big, T.max_len + cl + 512 * 6 + 16);      // staging area the largest third-list item could ask for
    if (cl > 0 && th_aln_class(T, cl, wdiv, wadd, wtol & 0xffff) == 2) atomicAdd(&C->n_item_long, T.pos_n - 1);     // (round 3's routing: what the long units add up to)
#pragma unroll
    for (int k = 0; k < TH_ITEM_COLS; ++k) rows[(long long)t * TH_ITEM_COLS + k] = c[k];
    atomicMax(&C->max_cl, cl);
}
__global__ void __launch_bounds__(256)
th_k_items_emit(const int *__restrict__ n_tasks, const ThTask *__restrict__ tasks, const int *__restrict__ cons_len, const int *__restrict__ status,
                int wdiv, int wadd, int wtol, const int *__restrict__ rows, const ThCnt *C, ThAlnItem *__restrict__ items, int *__restrict__ cc_off, th_job_t *__restrict__ jobs)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *n_tasks) return;
    const int st = status[t], cl = st ? 0 : cons_len[t];
    const ThTask T = tasks[t];
    const int cls = th_aln_class(T, cl, wdiv, wadd, wtol);
    const int base[3] = { 0, C->n_item_s, C->n_item_s + C->n_item_m };      // the three lists follow each other in one array
    int o = base[0] + rows[(long long)t * TH_ITEM_COLS];
    if (cls == 0) for (int u = 0; u < T.pos_n - 1; ++u) { ThAlnItem it = { t, u }; items[o++] = it; }
    { ThAlnItem l = { t, -1 }, r = { t, -2 }; items[o++] = l; items[o++] = r; }
    if (cls) {
        o = base[cls] + rows[(long long)t * TH_ITEM_COLS + cls];
        for (int u = 0; u < T.pos_n - 1; ++u) { ThAlnItem it = { t, u }; items[o++] = it; }
    }
    const int co = rows[(long long)t * TH_ITEM_COLS + 3];
    cc_off[t] = co;
    jobs[t].cons_off = co; jobs[t].cons_len = cl; jobs[t].status = st;
}

// every entry of an int array to one value (the score bounds of the unit alignments start as "none": a job whose POA never runs -- no scratch for it -- would otherwise
// hand th_k_align the PREVIOUS batch's bounds: ADVICE r5)
__global__ void __launch_bounds__(256) th_k_fill_i32(int *__restrict__ p, int n, int v) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v; }
__global__ void th_k_cnt_zero(ThCnt *C) { int *p = (int *)C; for (int i = threadIdx.x; i < (int)(sizeof(ThCnt) / 4); i += blockDim.x) p[i] = 0; }
// before a rerun of stages 2-3 / the job build: the flags that are being answered, and what the rerun accumulates again
__global__ void th_k_cnt_rerun(ThCnt *C, int redo_extract, int redo_part)
{
    if (threadIdx.x || blockIdx.x) return;
    C->err &= ~(TH_E_REC_OVERFLOW | TH_E_PART_OVERFLOW | TH_E_JOB_CAP);
    if (redo_extract) { C->rec = 0; C->q_xw = 0; C->q_xw2 = 0; }
    if (redo_part) { C->par_top = 0; C->q_part = 0; }
    C->need_n = 0; C->need_w = 0; C->need_l = 0; C->max_unit = 0; C->max_cons_cap = 0;
}
// sets single counters of the block between the segments of a rerun (slow paths only)
__global__ void th_k_cnt_set(int *p0, int v0, int *p1, int v1, int *p2, int v2, int *p3, int v3)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { if (p0) *p0 = v0; if (p1) *p1 = v1; if (p2) *p2 = v2; if (p3) *p3 = v3; }
}
