// th_dev.h -- device-side common definitions and wave-level helpers (gfx950: 64-lane waves).
#pragma once
#include "th_rt.h"

#define TH_WAVE 64
#define TH_NEG (-0x20000000)

// Parameters the kernels read (a device-friendly projection of mini_tandem_para, src/tidehunter.h:47-61)
struct ThDevPara {
    int k;                       // k-mer length (-k)
    int w;                       // -w: minimizer window (1 = every k-mer)
    int w_seq;                   // TH_MINIMIZER_SEQ=1: minimizer selection by the sequential replay only (the tests' cross-check)
    int part_full;               // TH_PART_FULL=1: the partition's bridging alignments on the full matrix only (round 3; A/B runs and the tests' cross-check)
    int sort_bitonic;            // TH_SORT_BITONIC=1: th_k_hits sorts with the bitonic network everywhere (no radix passes)
    int hpc;                     // -H: homopolymer-compressed k-mers (src/tandem_hit.c:46)
    uint32_t min_p, max_p;       // period limits, cast as at src/tandem_hit.c:235
    double one_minus_div;        // 1 - max_div (src/partition.c:202,252: the one double product on the path)
    int min_copy;
    int match, mismatch, go1, ge1, go2, ge2;   // POA scoring (src/abpoa_cons.c:12-28)
};

// one accepted chain (chain_t, src/tandem_chain.h:30-34) as the extract kernel emits it
struct ThChainRec { int read, cell_off, len, est_start, est_period, first_end, last_start, pad; };

// one consensus job = one run of >= min_copy+1 boundaries between -1 separators (src/gen_cons.c:192-205)
struct ThTask {
    int read;        // read index in the batch
    int pos_off;     // first boundary of the run in the compact par_pos array
    int pos_n;       // boundaries in the run (units = pos_n - 1)
    int unit_off;    // first POA unit in the unit table
    int n_units;     // units that pass the filter at src/abpoa_cons.c:43 (n_seqs)
    int sum_len, max_len;
    int cons_off;    // consensus / coverage output offset
    int cons_cap;
    int aln_off;     // first entry of this task in the per-unit iden_n array
    int narrow;      // 1: the job starts on the int16 POA variant (th_poa_job_is_narrow), 0: on the 32-bit one
    int noband;      // host, slow path: the job is being rerun without the adaptive band
};
#define TH_SLB_MIN_UNIT 2300           // th_k_poa computes the bounds for jobs whose longest unit has at least this many bases
#define TH_SLB_NONE (-0x40000000)      // no score bound for this unit alignment (th_k_poa: th_poa_unit_bounds; th_k_align)
struct ThUnit { int start, len, which, pad_; };   // start = index of the unit's first base inside the read; which = its index among the run's units (the alignment items' `which`: round 5)

// The pipeline's device counters: one block, zeroed by th_k_encode at the start of a batch.  Kernels get pointers to single fields.
struct ThCnt {
    int rec;                 // chain records in the arena (th_k_chain_extract*)
    int err;                 // error flags (TH_E_*)
    int q_poa_n, q_poa_w;    // work-queue heads: POA int16 variant / 32-bit variant
    int q_aln_s;             // ... alignment items, small staging area
    int par_top;             // top of the par_pos arena (th_k_partition)
    int q_xw, q_part;        // ... reads of the extraction waves, chains of the partition waves
    int min_fb;              // reads whose minimizer selection fell back to the sequential replay
    int q_aln_m, q_aln_b;    // ... alignment items, mid / big staging area
    int n_retry;             // jobs the POA gave back (status 1-4, 7): the host takes the slow path when this is not 0
    int n_failed;            // jobs that stay without a consensus (status 5, 6)
    int q_poa_l;             // ... POA relative-int16 variant (long units)
    int q_xw2;               // ... reads of the extraction waves, second instantiation (more than TH_XFAST_SMALL hits)
    int pad0[1];
    // job build (the first seven in the column order of th_k_jobs_count: the scan writes them in one go)
    int par_total, n_task, n_unit, cons_total, aln_total, n_task_n, n_task_l;
    int max_unit, max_cons_cap, pad1;
    unsigned long long need_n, need_w, need_l;       // POA scratch ints per wave the largest job of each variant asks for
    // item build
    int n_item_s, n_item_m, n_item_b, cc_total, max_cl, pad2[3];
    int n_defer, q_aln_d;    // unit alignments deferred to the strip-engine launch, and that launch's queue head
    int aln_wide[2];         // wide first bands (three registers per lane and more) tried / failed in the big-area alignment launch
    int max_need_big;        // bytes of staging area the largest item of the third alignment list could ask for
    int n_item_long;         // unit alignments of jobs too long for a two-register first band (th_aln_class: probed on the small variant, or the big variant's)
    int aln_first[5];        // unit alignments: first bands tried / not certified on the anti-diagonal engine, second bands tried / not certified; [4] flanks probed on a prefix (th_aln_long_flank) -- TH_TRACE prints them, the debug hook returns [4]
    int q_aln_pass[4];       // queue heads of the first alignment launch's earlier passes (longest flanks first: th_k_align, `pass_queue`)
    int pad3[9];
};
static_assert(sizeof(ThCnt) == 256, "counter block");
#define TH_E_REC_OVERFLOW 1       // chain record arena too small (rerun the extraction with a larger one)
#define TH_E_PART_OVERFLOW 2      // a par_pos region or the boundary buffer of the partition was too small
#define TH_E_ALN_BND 4            // boundary buffer of the alignment kernel too small
#define TH_E_MINIMIZER 8          // minimizer selection repeated a word or overflowed its buffer
#define TH_E_ALN_VERIFY 64        // TH_ALN_MODE=2: a banded alignment differs from the full matrix
#define TH_E_MSA_UNITS 128        // a job with fewer than two units ("Not enough sequences to perform msa.", src/abpoa_cons.c:61)
#define TH_E_ALN_BOUND 512        // th_k_align: a band sized by a lower bound found less than the bound (the bound was none: a bug in whoever supplied it)
#define TH_E_JOB_CAP 256          // job / unit / par_pos tables too small (rerun the build with larger ones)


// ---- packed int16 pairs (two DP cells per lane): v_pk_add_i16 clamp / v_pk_sub_i16 clamp / v_pk_max_i16 ----
#define TH_PK_NEG 0x80008000u            // (-32768, -32768): "minus infinity" in both halves
#define TH_PK_LO(x) ((int)(short)((x) & 0xffffu))
#define TH_PK_HI(x) ((int)(short)((uint32_t)(x) >> 16))
#define TH_PK(lo, hi) (((uint32_t)(lo) & 0xffffu) | ((uint32_t)(hi) << 16))
#define TH_PK_DUP(v) TH_PK((v), (v))

// The hardware primitives the kernels are written in -- TH_WAVE_FENCE (the point where data written by some lanes of a wave must be
// visible to the others), TH_READLANE / th_writelane, th_shr1 / th_shl1 / th_wave_incl_max / th_wave_incl_sum (DPP moves on the VALU:
// row_shr / row_bcast / wave_shr, a few cycles each, instead of ds_bpermute through the LDS crossbar at ~100 cycles per hop),
// TH_LOAD_L2, TH_VM_WAIT0, TH_UNI, TH_BALLOT, th_pk_* (packed int16) -- live in th_hw.h.
#include "th_hw.h"

// ---- wave helpers (every lane of the wave must call these; control flow around them is wave-uniform) ----
#define TH_INT_MIN (-0x7fffffff - 1)
__device__ __forceinline__ int th_wave_excl_max(int v, int seed) { return max(th_shr1(th_wave_incl_max(v), seed), seed); }
__device__ __forceinline__ int th_wave_max(int v) { return TH_READLANE(th_wave_incl_max(v), 63); }
__device__ __forceinline__ int th_wave_min(int v) { return -TH_READLANE(th_wave_incl_max(-v), 63); }
__device__ __forceinline__ int th_wave_sum(int v) { return TH_READLANE(th_wave_incl_sum(v), 63); }
// lane l <- lane l+off for a wave-uniform offset (fill where the source lane does not exist)
__device__ __forceinline__ int th_lane_shift(int v, int off, int fill)
{
    if (off == 0) return v;
    if (off == -1) return th_shr1(v, fill);
    if (off == 1) return th_shl1(v, fill);
    if (off == 2) return th_shl1(th_shl1(v, fill), fill);
    const int src = TH_LANE() + off;
    const int o = __shfl(v, src & 63);
    return (src >= 0 && src < TH_WAVE) ? o : fill;
}
// next index of a shared work queue, wave-uniform.  Every lane takes part in the atomic (lane 0 adds 1, the
// others 0), so there is no divergent region around it: with the add under `if (lane == 0)` hipcc 7.2
// restructured the surrounding persistent loop into one that never fetched a second item.
__device__ __forceinline__ int th_queue_pop(int *queue)
{
    const int v = atomicAdd(queue, TH_LANE() == 0 ? 1 : 0);
    return TH_READLANE(v, 0);
}
__device__ __forceinline__ int th_ctz64(unsigned long long m) { return m ? __ffsll((long long)m) - 1 : 64; }
__device__ __forceinline__ int th_msb64(unsigned long long m) { return 63 - __clzll((long long)m); }   // m != 0
// the same for masks known to be non-zero: bare s_ff1 / s_flbit, no zero fix-ups
__device__ __forceinline__ int th_ctz64nz(unsigned long long m) { return __builtin_ctzll(m); }
__device__ __forceinline__ int th_msb64nz(unsigned long long m) { return 63 - __builtin_clzll(m); }

// ---- workgroup bitonic sort of N = 2^k words in LDS (or HBM), ascending or descending; ends with a barrier ----
// A compare-exchange stage costs a round trip through the buffer and a barrier, so stages are grouped: a thread loads 32
// words into registers, runs up to five consecutive stages there and stores them.  The grouped pass with unit u takes the words
// base + x u (x = 0..31; consecutive lanes take consecutive bases, so the strided LDS accesses do not collide) and covers
// the strides 16 u .. u that are <= the current one: u = 32 for the strides 512 .. 32, u = 1 for 16 .. 1.  A merge of size
// 2^k thus costs max(k - 10, 0) single passes + 2 grouped ones instead of k (33 passes instead of 105 at 16384 words).
template <typename W, bool DESC, int BLOCK>
__device__ __forceinline__ void th_bitonic_sort(W *buf, int N, int tid)
{
    auto reg_pass = [&](int u, int size, int stride) {
        for (int w = tid; w < (N >> 5); w += BLOCK) {
            W v[32];
            const int base = (w / u) * 32 * u + (w & (u - 1));
#pragma unroll
            for (int x = 0; x < 32; ++x) v[x] = buf[base + x * u];
#pragma unroll
            for (int st = 16; st > 0; st >>= 1) {
                if (st * u <= stride) {
#pragma unroll
                    for (int x = 0; x < 32; ++x) {
                        if ((x & st) == 0) {
                            const bool first = (((base + x * u) & size) == 0) != DESC;      // the smaller word goes first
                            const W a = v[x], c = v[x + st];
                            const W mn = a < c ? a : c, mx = a < c ? c : a;
                            v[x] = first ? mn : mx; v[x + st] = first ? mx : mn;
                        }
                    }
                }
            }
#pragma unroll
            for (int x = 0; x < 32; ++x) buf[base + x * u] = v[x];
        }
        __syncthreads();
    };
    const int top = N >= 1024 ? 512 : N >= 32 ? 16 : 0;      // largest stride the grouped passes take
    for (int size = 2; size <= N; size <<= 1) {
        int stride = size >> 1;
        for (; stride > top; stride >>= 1) {
            for (int t = tid; t < (N >> 1); t += BLOCK) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool first = ((lo & size) == 0) != DESC;
                const W a = buf[lo], c = buf[hi];
                if ((a > c) == first) { buf[lo] = c; buf[hi] = a; }
            }
            __syncthreads();
        }
        if (N >= 1024 && stride >= 32) { reg_pass(32, size, stride); stride = 16; }
        if (N >= 32 && stride >= 1) reg_pass(1, size, stride);
    }
}

// ---- workgroup LSD radix sort of 32-bit words in LDS by a key field, stable; ends with a barrier ----
// For words whose order among equal keys is already right (th_k_hits: key << PB | position, written in position order), sorting by the key
// FIELD alone with a stable sort gives the fully sorted sequence -- at 2 passes of 8-bit digits for 16 key bits, where the bitonic network
// above makes 33 passes over the same buffer (105 compare-exchange stages at 16384 words) and fights the LDS banks at every stride.
// Per pass: every wave owns a contiguous 1/WAVES of the array and walks it 64 words at a time (word = chunk base + step * 64 + lane: the
// loads are conflict-free and land in registers, so the scatter can go back into the same buffer); the digits are counted per wave
// (LDS atomics), a bin-major scan over (digit, wave) turns the counts into start offsets, and the scatter ranks the lanes of a step that
// share a digit among themselves with eight ballots (match-any) -- lane order = array order, which is what keeps the sort stable.
// Words equal to `none` sink to the end (an extra bin of the last pass).  N: a power of two, 64 * WAVES <= N <= 32 * 64 * WAVES.
// hist: WAVES x (TH_RDX_BINS + 1) words of LDS; tot: 16 words.
#define TH_RDX_BINS 256
#define TH_RDX_STEPS 32
template <int BLOCK>
__device__ __forceinline__ void th_radix_sort_field(uint32_t *buf, int N, int shift, int kbits, uint32_t none, uint32_t (*hist)[TH_RDX_BINS + 1], int *tot, int tid)
{
    constexpr int WAVES = BLOCK / TH_WAVE;
    const int lane = tid & 63, wave = tid >> 6;
    const int chunk = N / WAVES, steps = chunk >> 6;
    uint32_t *my = buf + wave * chunk + lane;
    const int passes = (kbits + 7) >> 3;
    for (int p = 0; p < passes; ++p) {
        const int sh = shift + 8 * p; const bool lastp = p == passes - 1;
        const uint32_t dmask = lastp ? (1u << (kbits - 8 * p)) - 1u : 0xffu;
        for (int x = tid; x < WAVES * (TH_RDX_BINS + 1); x += BLOCK) hist[0][x] = 0;
        __syncthreads();
        uint32_t v[TH_RDX_STEPS];
#pragma unroll
        for (int s_ = 0; s_ < TH_RDX_STEPS; ++s_) {
            v[s_] = none;
            if (s_ < steps) {
                v[s_] = my[s_ << 6];
                const uint32_t d = (lastp && v[s_] == none) ? TH_RDX_BINS : ((v[s_] >> sh) & dmask);
                atomicAdd(&hist[wave][d], 1u);
            }
        }
        __syncthreads();
        {   // counts -> start offsets, bin-major over (digit, wave); the waves that hold bins 0 .. 255 scan their totals, bin 256 follows all of them
            uint32_t t_ = 0;
            if (tid < TH_RDX_BINS) for (int w = 0; w < WAVES; ++w) t_ += hist[w][tid];
            const int inc = th_wave_incl_sum((int)t_);
            if (tid < TH_RDX_BINS && lane == 63) tot[wave] = inc;
            __syncthreads();
            if (tid < TH_RDX_BINS) {
                uint32_t acc = (uint32_t)inc - t_;
                for (int w = 0; w < wave; ++w) acc += (uint32_t)tot[w];
                for (int w = 0; w < WAVES; ++w) { const uint32_t c = hist[w][tid]; hist[w][tid] = acc; acc += c; }
            }
            if (tid == TH_RDX_BINS % BLOCK) {          // the sentinels' bin: behind every word (a thread of its own in a block of 512, thread 0's second bin in one of 256)
                uint32_t acc = 0;
                for (int w = 0; w < TH_RDX_BINS / TH_WAVE; ++w) acc += (uint32_t)tot[w];
                for (int w = 0; w < WAVES; ++w) { const uint32_t c = hist[w][TH_RDX_BINS]; hist[w][TH_RDX_BINS] = acc; acc += c; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s_ = 0; s_ < TH_RDX_STEPS; ++s_) {
            if (s_ < steps) {
                const uint32_t d = (lastp && v[s_] == none) ? TH_RDX_BINS : ((v[s_] >> sh) & dmask);
                unsigned long long mm = ~0ull;                      // lanes of this step with my digit
#pragma unroll
                for (int b = 0; b < 9; ++b) {
                    if (b < 8 || lastp) {
                        const bool bit = (d >> b) & 1u;
                        const unsigned long long m = __ballot(bit);
                        mm &= bit ? m : ~m;
                    }
                }
                const int rank = th_mbcnt(mm), cnt = __popcll(mm);
                const uint32_t base = hist[wave][d];
                TH_WAVE_FENCE();                                    // every lane has read its group's offset before the group's last lane moves it on
                if (rank == cnt - 1) hist[wave][d] = base + (uint32_t)cnt;
                buf[base + (uint32_t)rank] = v[s_];
                TH_WAVE_FENCE();
            }
        }
        __syncthreads();
    }
}

// (a & m) | (b & ~m): v_bfi_b32
__device__ __forceinline__ uint32_t th_sel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
