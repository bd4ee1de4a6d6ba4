// th_k_align.hip -- stage 4 follow-up: every unit against the consensus (aveMatch) and the consensus
// extended into both flanks (start / end / fractional copy number), gfx950.
//
// Replaces the ksw2 calls of seqs_msa(), src/gen_cons.c:209-223:
//   ksw2_global(unit, cons) per unit, ksw2_left_ext(cons, read[0..first]), ksw2_right_ext(cons, read[last+1..]).
// One alignment per wavefront (th_ksw.h), waves loop over the item list.
#include "th_dev.h"
#include "th_ksw.h"

#define TH_ALN_BLOCK 64
// The two-register variant is compiled for THREE waves per SIMD (at most 168 vector registers, which the allocator hands out in blocks of eight:
// 169 registers -- what the kernel asked for when the packed strip engines of round 4 were added -- meant two waves and 8 % on every shape),
// the six-register variants for two.
#ifndef TH_ALN_OCC
#define TH_ALN_OCC __attribute__((amdgpu_waves_per_eu(NPMAX <= 2 ? 3 : 2, NPMAX <= 2 ? 3 : 2)))
#endif

// Compacts the consensus strings (and, on request, their per-base coverage) from their capacity-sized slots to the
// actual lengths, so that the device-to-host copy moves ~1.1 KB per read instead of the 3 x unit-length slot.
__global__ void th_k_gather_cons(int n_tasks, const ThTask *__restrict__ tasks, const int *__restrict__ len, const int *__restrict__ dst_off,
                                 const uint8_t *__restrict__ cons, const int *__restrict__ cov, uint8_t *__restrict__ cons_out, int *__restrict__ cov_out)
{
    const int t = blockIdx.x;
    if (t >= n_tasks) return;
    const int n = len[t], so = tasks[t].cons_off, o = dst_off[t];
    for (int i = threadIdx.x; i < n; i += blockDim.x) { cons_out[o + i] = cons[so + i]; if (cov_out) cov_out[o + i] = cov[so + i]; }
}

// Anti-diagonal band engine (th_ksw_band_ad_wave): NP registers per lane cover 256 NP diagonals.  th_ad_pick returns the
// smallest NP <= npmax whose band reaches `want` diagonals beyond both 0 and D, or 0 when none does or the staging area is too small.
__device__ __forceinline__ int th_ad_pick(int qlen, int tlen, int want, int lds_bytes, int npmax)
{
    const int D = tlen - qlen, aD = D < 0 ? -D : D;
    if (qlen + tlen >= 30000) return 0;
    for (int np = 1; np <= npmax; ++np)
        if ((256 * np - 2 - aD) / 2 >= want) return qlen + tlen + 512 * np + 16 <= lds_bytes ? np : 0;
    return 0;
}
template <int NP>
__device__ __forceinline__ int th_ad_run(bool has_n, const uint8_t *st, const uint8_t *sq, int qlen, int tlen, int lo, int hi, int *score)
{
    return has_n ? th_ksw_band_ad_wave<NP, true>(st, sq, qlen, tlen, lo, hi, score) : th_ksw_band_ad_wave<NP, false>(st, sq, qlen, tlen, lo, hi, score);
}
template <int NPMAX>
__device__ __forceinline__ int th_ad_align(int np, const uint8_t *__restrict__ q, int qlen, const uint8_t *__restrict__ t, int tlen, uint8_t *s_seq, int *score)
{
    // (the lengths come out of per-lane loads: said to be wave-uniform here, the engine's loop counters and phase limits live in scalar registers and its loops are
    // scalar loops -- two vector instructions of a pair of anti-diagonals' 77 were the loop's own counter and compare: round 5)
    np = TH_UNI(np); qlen = TH_UNI(qlen); tlen = TH_UNI(tlen);
    const int D = tlen - qlen, aD = D < 0 ? -D : D;
    const int W = (256 * np - 2 - aD) / 2;
    int lo = min(0, D) - W; lo -= lo & 1;                   // even; the band is as wide as the registers allow
    const int hi = lo + 256 * np - 1;
    const uint8_t *st, *sq;
    const bool has_n = th_ad_stage(q, qlen, t, tlen, 128 * np, s_seq, &st, &sq);
    int v = -1;
    if (np == 1) v = th_ad_run<1>(has_n, st, sq, qlen, tlen, lo, hi, score);
    else if (np == 2) v = th_ad_run<2>(has_n, st, sq, qlen, tlen, lo, hi, score);
    else if (NPMAX >= 3 && np == 3) v = th_ad_run<(NPMAX >= 3 ? 3 : 1)>(has_n, st, sq, qlen, tlen, lo, hi, score);
    else if (NPMAX >= 4 && np == 4) v = th_ad_run<(NPMAX >= 4 ? 4 : 1)>(has_n, st, sq, qlen, tlen, lo, hi, score);
    else if (NPMAX >= 5 && np == 5) v = th_ad_run<(NPMAX >= 5 ? 5 : 1)>(has_n, st, sq, qlen, tlen, lo, hi, score);
    else if (NPMAX >= 6 && np == 6) v = th_ad_run<(NPMAX >= 6 ? 6 : 1)>(has_n, st, sq, qlen, tlen, lo, hi, score);
    TH_WAVE_FENCE();
    return v;
}

// Flank extension on the anti-diagonal engine (th_ksw_ext_ad_wave); false when the result could not be certified.
template <int NP>
__device__ __forceinline__ bool th_adx_run(bool has_n, const uint8_t *st, const uint8_t *sq, int qlen, int tlen, int *mq, int *mt, int *bh)
{
    return has_n ? th_ksw_ext_ad_wave<NP, true>(st, sq, qlen, tlen, mq, mt, bh) : th_ksw_ext_ad_wave<NP, false>(st, sq, qlen, tlen, mq, mt, bh);
}
template <int NPXMAX>
__device__ __forceinline__ bool th_adx_align(int np, const uint8_t *__restrict__ q, int qs, int qlen, const uint8_t *__restrict__ t, int ts, int tlen, uint8_t *s_seq, int *mq, int *mt, int *bh)
{
    np = TH_UNI(np); qlen = TH_UNI(qlen); tlen = TH_UNI(tlen);
    const uint8_t *st, *sq;
    const bool has_n = th_ad_stage(q, qlen, t, tlen, 128 * np, s_seq, &st, &sq, qs, ts);
    bool ok = false;
    if (np == 1) ok = th_adx_run<1>(has_n, st, sq, qlen, tlen, mq, mt, bh);
    else if (np == 2) ok = th_adx_run<2>(has_n, st, sq, qlen, tlen, mq, mt, bh);
    else if (NPXMAX >= 3 && np == 3) ok = th_adx_run<(NPXMAX >= 3 ? 3 : 1)>(has_n, st, sq, qlen, tlen, mq, mt, bh);
    else if (NPXMAX >= 4 && np == 4) ok = th_adx_run<(NPXMAX >= 4 ? 4 : 1)>(has_n, st, sq, qlen, tlen, mq, mt, bh);
    TH_WAVE_FENCE();
    return ok;
}

// A flank that does not fit the staging area next to the consensus even with ONE register's band (config 4: a first unit boundary found a period late leaves 7000-10 000
// flank bases for a 5000-base consensus) used to go to the FULL matrix unprobed: 80 of config 4's 40 010 extensions, 100-390 M clocks each where the others take 20 M --
// the last 80 of the first launch's 213 ms belonged to them (-DTH_ALN_PROF=1, round 5).  It is probed on the PREFIX that fits: cell (i, j) of an extension depends on the
// prefixes i and j alone, so every cell of the prefix's matrix has the same value in the whole one, and a banded maximum over the prefix is a lower bound of the whole
// maximum -- all the pruned strips need (such a band certifies the prefix, not the flank: only its maximum is used).  Two registers: four found the same bound (the
// alignment leaves either band after ~2000 bases at 15 %), and the four-register engine inlined here spilled -- 128 bytes more of stack per lane for the whole KERNEL,
// which cost config 5, where no such flank exists, 14 of its 405 ms.  A probe that finds nothing positive -- a flank that does not continue the repeat -- still prunes: 1 bounds the
// maximum from below whenever a positive cell exists, and when none does the pruned sweep finds none either (ksw2 reports (-1, -1): its maximum starts at 0 and only a larger
// H moves it, ksw2_extz2_sse.c), so max(bh, 1) is exact for every pair; H of such a flank falls by ~0.6 per base, and a cell whose H plus the diagonal steps left is
// below 1 is dead.  Out of line: the common path keeps its registers.
template <int LDSB>
__device__ TH_NOINLINE void th_aln_long_flank(const uint8_t *__restrict__ qp, int st, int cl, const uint8_t *__restrict__ tp, int tlen, uint8_t *s_seq, int *__restrict__ mybnd, int (*s_in)[TH_WAVE][8],
                                              bool pkstrip, int *mq, int *mt, int *bh_out)
{
    const int tl_p = min(tlen, LDSB - cl - 16 - 512 * 2);
    int a_ = -1, b_ = -1, bh = 0;
    (void)th_adx_align<2>(2, qp, st, cl, tp, st, tl_p, s_seq, &a_, &b_, &bh);
    const int slb = max(bh, 1);
    int W = 512;
    while (W < cl + tlen && th_ext_bound_w(W, cl, tlen) >= slb) W += 64;
    if (pkstrip && cl + tlen <= TH_KSW_PK_MAXLEN) th_ksw_ext_band_pk_wave(qp, st, cl, tp, st, tlen, W, slb, mybnd, s_in, mq, mt);
    else th_ksw_ext_band_wave(qp, st, cl, tp, st, tlen, W, slb, mybnd, s_in, mq, mt);
    *bh_out = bh;
}

// mode: 0 strip engine only (th_ksw_global_band_wave), 1 anti-diagonal engine first (default), 2 both and compare (err bit 64 on a
// difference), 3 strip engine for the deferred items of the launch before (see `defer`; item_s1: their banded scores).
// n_items_ptr: the item count when it only exists on the device (the deferred list).
// defer / defer_n / defer_s1: a unit alignment that the anti-diagonal engine could not certify is not finished here but appended to this
// list with its banded score, and a second launch of the small-staging-area variant takes it in mode 3: what is left -- a wider certified
// band or the full matrix -- runs on the strip engine, which needs 4 KB of LDS, and at this variant's 24 KB and 250 registers it would
// run at one or two waves per SIMD (config 4: 5000 x 5000 pairs at 15 % error were 470 of the 849 ms of a step that way).
#define TH_ALN_NO_S1 (-0x40000000)
#if defined(TH_ALN_PROF)
// -DTH_ALN_PROF=1 (tools/build_variant.sh; never the product): how evenly a launch's dynamic queue keeps its waves busy -- per wave that took an item, the ticks from its
// start to its exit: [0] waves, [1] shortest, [2] longest, [3] sum >> 10, [4] items; printed and reset by th_pipeline.hip after every launch
__device__ unsigned long long th_aln_prof[40] = { 0, ~0ull };      // [8 + b]: waves whose ticks >> 25 == b (histogram, last bucket open); [5] ticks of the costliest item, [6] its index
// round 6: where a launch's time goes BY CLASS of item -- [class][0] items, [1] ticks (a wave works on one item at a time: the classes' ticks sum to the waves' busy time),
// [2] cells the register-band engine computed (anti-diagonals x 128 x registers, every band run on the item), [3] cells of the band the item's FINAL score certifies
// (anti-diagonals / 2 x its diagonals: what any exact banded method must at least look at).  Classes: 0 / 1 unit alignment certified on its first band of one / two registers,
// 2 on a second register band, 3 handed to the strip launch (time until then), 4 strip engines or full matrix here, 5 / 6 left extension certified on its first band / not,
// 7 / 8 right extension likewise, 9 nothing to align
__device__ unsigned long long th_aln_cls[12][4];
#endif
// a flank extension that cannot be staged next to its consensus even with ONE register's band: the class the longest-first walk takes first and th_aln_long_flank serves
// (ONE predicate for both sites: ADVICE r5)
template <int LDSB> __device__ __forceinline__ bool th_aln_unstaged_flank(int cl, int tlen) { return cl + tlen + 512 + 16 > LDSB; }
template <int LDSB, int NPMAX, bool LPT = false>      // LPT: the list is walked longest flanks first (`pass_queue`; an instantiation of its own: the plain walk keeps its registers)
__global__ void __launch_bounds__(TH_ALN_BLOCK) TH_ALN_OCC
th_k_align(const ThAlnItem *__restrict__ items, int n_items, const ThTask *__restrict__ tasks, const int *__restrict__ pos,
           const uint32_t *__restrict__ roff, const uint8_t *__restrict__ bseq,
           const uint8_t *__restrict__ cons, const int *__restrict__ cons_len,
           int *__restrict__ bnd, long long bnd_stride, int *__restrict__ iden, int *__restrict__ ext, int *__restrict__ err, int *__restrict__ queue,
           int wdiv, int wadd, int wtol, int mode, int xpct, const int *__restrict__ n_items_ptr, ThAlnItem *__restrict__ defer, int *__restrict__ defer_n, int *__restrict__ wide_stat,
           const int *__restrict__ item_s1, int *__restrict__ defer_s1, int *__restrict__ aln_stat = nullptr,
           const int *__restrict__ unit_slb = nullptr /* round 5: per unit alignment (T.aln_off + which) a lower bound of its ksw2 score from the POA's own alignment of the unit, or TH_SLB_NONE */,
           int *__restrict__ pass_queue = nullptr /* round 5: four more queue heads -- the list is walked five times, LONGEST FLANKS FIRST (see the loop) -- or null: one walk in list order */)
{
    // one LDS area, used either as the left-border inbox of the strip sweep (two 64-row blocks, 4 KB) or as the staged bases
    // of the anti-diagonal engine (never both at once)
    static_assert(LDSB >= 2 * TH_WAVE * 8 * 4, "the staging area doubles as the strip engine's inbox");
    __shared__ __attribute__((aligned(16))) uint8_t s_seq[LDSB];
    int (*s_in)[TH_WAVE][8] = (int (*)[TH_WAVE][8])s_seq;
    const int lane = TH_LANE();
    const int wave = blockIdx.x;
    int *mybnd = bnd + (long long)wave * bnd_stride;
    // registers per lane of the flank extensions' PROBE (upper bits of xpct; 1 by default): a long flank no register band is expected to certify is
    // probed for the lower bound that prunes its strip sweep -- the repeat's continuation hugs the main diagonal, so the narrowest band finds
    // (nearly) the true maximum at a quarter of the four-register probe of round 3
    // (taken out of the ONE word where they are used -- cold places all -- instead of four values held across the item loop: this kernel's scalar registers are full, and
    // what does not fit is spilled to vector lanes; config 5's 392 ms moved by 15 with one more such value, round 5)
    const int xword = xpct;
#define xprobe (min(4, max(1, ((xword >> 8) & 7))))
#define longflank (!((xword >> 12) & 1))       /* bit 12 (TH_ALN_LONGFLANK=0): flanks too long for the staging area go to the full matrix unprobed, as before round 5 (A/B runs) */
#define pkstrip (!((xword >> 11) & 1))         /* bit 11 (TH_ALN_PKSTRIP=0): the one-column strip engines only (A/B runs, the tests' cross-check) */
#define xpct (xword & 0xff)
    if (n_items_ptr) n_items = TH_UNI(*n_items_ptr);
#if defined(TH_ALN_PROF)
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime(); int prof_items = 0;
#endif
    // Longest first (round 5).  A launch with few items per wave ends when its unluckiest wave does: config 4's first launch -- 40 010 flank extensions of 1 to 5000 bases,
    // 13 per wave, in job order -- kept its waves busy 0.62 of the time of the longest one (the -DTH_ALN_PROF=1 build), 80 of its 213 ms.  With `pass_queue` the list is
    // walked five times, each walk through a queue head of its own and taking only its class: flanks too long for the staging area (probed on a prefix, finished on the
    // pruned strips: the costliest items by far), flanks of >= 3072 bases, >= 1536, >= 768, then everything else; a wave that
    // pops an item of another class drops it (three loads).  The host asks for this when a launch has fewer than 64 items per wave; where items are plenty (the headline
    // shape: 358 per wave) the order does not matter and the extra pops would only cost.
    constexpr int n_pass = LPT ? 5 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
    int *const q_ = (!LPT || pass == n_pass - 1) ? queue : pass_queue + pass;
    for (;;) {
        const int it = th_queue_pop(q_);   // dynamic work queue: items range from a short extension to a full matrix
        if (it >= n_items) break;
        const ThAlnItem I = items[it];
        const ThTask T = tasks[I.task];
        const int cl = cons_len[I.task];
        const uint8_t *rb = bseq + roff[T.read];
        const int L = (int)(roff[T.read + 1] - roff[T.read]);
        const uint8_t *cs = cons + T.cons_off;
        const int *ps = pos + T.pos_off;
        if (LPT) {
            int cls = 4;
            if (I.which < 0) { const int tl = I.which == -1 ? ps[0] + 1 : L - ps[T.pos_n - 1] - 1; cls = th_aln_unstaged_flank<LDSB>(cl, tl) ? 0 : tl >= 3072 ? 1 : tl >= 1536 ? 2 : tl >= 768 ? 3 : 4; }
            if (TH_UNI(cls) != pass) continue;
        }
#if defined(TH_ALN_PROF)
        ++prof_items;
        const unsigned long long prof_i0 = __builtin_amdgcn_s_memtime();
        unsigned long long prof_desc = 0;      // flank items: tlen | cl << 16 | max(bh, 0) << 32 | path << 60 (1 certified band, 2 pruned strips, 3 full matrix; +4: prefix probe)
        int prof_cls = 9; unsigned long long prof_cells = 0, prof_need = 0;
        struct ProfItem { unsigned long long t0; int it, lane; const unsigned long long *desc; const int *cls; const unsigned long long *cells, *need;
                          __device__ ~ProfItem() { if (lane == 0) { const unsigned long long d = __builtin_amdgcn_s_memtime() - t0; if (atomicMax(&th_aln_prof[5], d) < d) { th_aln_prof[6] = (unsigned long long)it; th_aln_prof[7] = *desc; }
                                                                   atomicAdd(&th_aln_cls[*cls][0], 1ull); atomicAdd(&th_aln_cls[*cls][1], d); atomicAdd(&th_aln_cls[*cls][2], *cells); atomicAdd(&th_aln_cls[*cls][3], *need); } } } prof_item_{ prof_i0, it, lane, &prof_desc, &prof_cls, &prof_cells, &prof_need };
#endif
        if (I.which >= 0) {
            int v = 0;
            if (cl > 0) {
                const int start = ps[I.which], len = ps[I.which + 1] - start;
                if ((long long)((cl + 63) & ~63) * 4 > bnd_stride) { if (lane == 0) atomicOr(err, 4); }
                else if (len > 0) {
                    // banded pass with an exactness certificate first; the full matrix only when it cannot be certified
                    // first band (cons_len / 10 + 24 by default): wide enough to be certified for units up to ~11 % away from the consensus;
                    // more divergent units fail the certificate and get a band sized from their banded score.  Measured on config 3
                    // (10 % read error) with the strip engine, alignment ms per 100 000 reads: /8+32 393, /9+24 367, /10+24 353, /11+24 358, /12+24 389
                    const int W = cl / wdiv + wadd, D = cl - len, aD = D < 0 ? -D : D;
                    const uint8_t *qp = rb + start + 1;
                    v = -1;
                    int s1 = 0; bool have_s1 = false;
                    // anti-diagonal engine: its band is as wide as its registers (>= W - wtol), so it also serves matrices narrower than the band
                    int np = (mode == 1 || mode == 2) ? th_ad_pick(len, cl, max(W - wtol, 0), LDSB, NPMAX) : 0;
                    // no band of this variant's registers reaches the first band asked for (a long unit on the two-register variant: th_aln_class):
                    // ONE register as a probe -- its banded score sizes the band that can be certified and prunes the strip sweep that follows
                    // Round 5 (configs 4 / 5): a LONG unit -- one whose first band would be a one-register probe or three registers and more -- starts from the bound the POA
                    // left for it (th_poa_unit_bounds: the score of the unit's own path through the graph against the consensus columns, a valid global alignment):
                    // the band that is certified for any optimum >= that bound is run at once -- on the registers if they reach it, else on the pruned strips with
                    // the bound as their lower bound -- instead of a probe whose only product was a (worse) bound.  Exactness does not rest on the bound: the register
                    // band carries its own certificate, and a strip sweep whose banded score comes out BELOW the bound (impossible for a valid one) is redone in full.
                    const int slb0 = (unit_slb != nullptr && (mode == 1 || mode == 2)) ? TH_UNI(unit_slb[T.aln_off + I.which]) : TH_SLB_NONE;
                    const bool from_slb = slb0 != TH_SLB_NONE && (np == 0 || np >= 3);
                    if (from_slb) {
                        const long long g2 = ((long long)len + cl - 2LL * slb0 - 4 * TH_KQ) / (1 + 2 * TH_KE) + 2;
                        const long long W2 = (g2 - 2 - aD + 1) / 2 + 8;
                        np = W2 < 4096 ? th_ad_pick(len, cl, (int)W2, LDSB, NPMAX) : 0;
                        s1 = slb0; have_s1 = true;
                    }
                    if (!from_slb && np == 0 && (mode == 1 || mode == 2) && NPMAX <= 2 && th_ad_pick(len, cl, 0, LDSB, NPMAX) == 1) np = 1;
                    // A first band of three registers per lane or more (units of several thousand bases) that keeps failing its certificate in
                    // this launch -- pairs as divergent as config 4's, 15 % error: no band short of the matrix can be certified -- is probed
                    // with ONE register first: the probe's banded score sizes the band that could be certified, and the pair goes on to the
                    // full matrix for a sixth of the wide pass it would have wasted.  The launch counts its wide first bands and their
                    // failures (wide_stat[0], [1]); which engine computes an alignment never changes its result.
                    const bool wide_first = !from_slb && np >= 3 && wide_stat != nullptr;
                    if (wide_first) {
                        const int tries = TH_LOAD_L2(&wide_stat[0]), fails = TH_LOAD_L2(&wide_stat[1]);
                        if (TH_UNI(tries) >= 8 && 2 * TH_UNI(fails) > TH_UNI(tries) && th_ad_pick(len, cl, 0, LDSB, NPMAX) == 1) np = 1;
                    }
                    const int np_first = np;
                    const bool ad_first = np != 0;
                    // half-width of the widest band actually run on this pair so far (the follow-up compared W2 with the first band ASKED for, W, although a
                    // one-register probe may have run in its place -- a 5000-base unit 5 % from its consensus has W2 ~ 300 < W = 524 and went to the 32-bit full matrix
                    // instead of a pruned band of W2); -1: nothing ran here (mode 3: the launch before did, and did not certify, so its band was narrower than W2)
                    int w_tried = -1;
                    while (np) {
                        v = th_ad_align<NPMAX>(np, qp, len, cs, cl, s_seq, &s1); have_s1 = true; w_tried = max(w_tried, (256 * np - 2 - aD) / 2);
#if defined(TH_ALN_PROF)
                        prof_cells += (unsigned long long)(len + cl) * 128ull * (unsigned long long)np;
                        if (v >= 0) { prof_cls = np != np_first ? 2 : np == 1 ? 0 : 1;
                                      const long long g_ = ((long long)len + cl - 2LL * s1 - 4 * TH_KQ) / (1 + 2 * TH_KE) + 2, w_ = (g_ - 2 - aD + 1) / 2;
                                      prof_need = (unsigned long long)((len + cl) / 2) * (unsigned long long)(2 * max(w_, 0LL) + aD + 1); }
#endif
                        if (from_slb && s1 < slb0 && lane == 0) atomicOr(err, TH_E_ALN_BOUND);      // (the band was sized to hold a path of score slb0: a banded score below it says the POA's bound was none -- never seen; the tests would)
                        if (wide_first && np == np_first && np >= 3 && lane == 0) { atomicAdd(&wide_stat[0], 1); if (v < 0) atomicAdd(&wide_stat[1], 1); }
                        if (aln_stat && lane == 0) { const int o_ = np == np_first ? 0 : 2; atomicAdd(&aln_stat[o_], 1); if (v < 0) atomicAdd(&aln_stat[o_ + 1], 1); }
                        if (v >= 0) break;
                        // not certified: the banded score s1 bounds the optimum from below, and a band whose Gmin exceeds
                        // (m + n - 2 s1 - 4q) / (1 + 2e) is certified for any optimum >= s1
                        const long long g2 = ((long long)len + cl - 2LL * s1 - 4 * TH_KQ) / (1 + 2 * TH_KE) + 2;
                        const long long W2 = (g2 - 2 - aD + 1) / 2 + 8;
                        const int np2 = W2 < 4096 ? th_ad_pick(len, cl, (int)W2, LDSB, NPMAX) : 0;
                        np = np2 > np ? np2 : 0;
                    }
                    // (the strip engines: two columns per lane on packed int16 where the scores fit -- round 4 -- else one column per lane in 32 bits)
                    const bool pkg = pkstrip && len + cl <= TH_KSW_PK_MAXLEN;
                    if (!ad_first && !from_slb && mode != 3 && 64 + 2 * W + aD < (cl * 3) / 4) { v = pkg ? th_ksw_global_band_pk_wave<false>(qp, len, cs, cl, W, mybnd, s_in, &s1) : th_ksw_global_band_wave(qp, len, cs, cl, W, mybnd, s_in, &s1); have_s1 = true; w_tried = max(w_tried, W); }
                    if (mode == 3) { s1 = item_s1[it]; have_s1 = s1 != TH_ALN_NO_S1; }       // deferred by the launch before, with its banded score
                    if (v < 0 && defer && mode == 1) {          // what is left is strip-engine work (a wider certified band, the full matrix): on that engine's own launch
                        int slot = 0;
                        if (lane == 0) { slot = atomicAdd(defer_n, 1); defer[slot] = I; defer_s1[slot] = have_s1 ? s1 : TH_ALN_NO_S1; }
#if defined(TH_ALN_PROF)
                        prof_cls = 3;
#endif
                        continue;
                    }
#if defined(TH_ALN_PROF)
                    if (v < 0 || prof_cls == 9) prof_cls = 4;
#endif
                    if (v < 0 && have_s1) {
                        // strip engine for the second, wider certified band (more divergent than ~12 %)
                        const long long g2 = ((long long)len + cl - 2LL * s1 - 4 * TH_KQ) / (1 + 2 * TH_KE) + 2;
                        const long long W2 = (g2 - 2 - aD + 1) / 2 + 8;
                        // (with the banded score as a lower bound the engine also drops every cell no path of that score can pass through:
                        // never more work than the matrix, so no area test)
                        if (W2 > w_tried && W2 < 30000) {
                            const int lb = s1; v = pkg ? th_ksw_global_band_pk_wave<false>(qp, len, cs, cl, (int)W2, mybnd, s_in, &s1, lb) : th_ksw_global_band_wave(qp, len, cs, cl, (int)W2, mybnd, s_in, &s1, lb);
                            // (a path of score lb fits the band its score sizes, so the sweep finds at least lb -- unless lb was no lower bound at all: checked
                            // here for every bound, the POA's included, loudly, and the pair redone in full)
                            if (s1 < lb) { v = -1; if (lane == 0) atomicOr(err, TH_E_ALN_BOUND); }
                        }
                    }
                    if (v < 0) v = th_ksw_global_wave<false>(qp, len, cs, cl, 0, mybnd, nullptr);
                    if (mode == 2) {
                        const int v0 = th_ksw_global_wave<false>(qp, len, cs, cl, 0, mybnd, nullptr);
                        if (v0 != v && lane == 0) atomicOr(err, 64);
                    }
                }
            }
            if (lane == 0) iden[T.aln_off + I.which] = v;
        } else {
            int mq = -1, mt = -1;
            if (cl > 0) {
                int tlen; const uint8_t *tp, *qp; int st;
                if (I.which == -1) { tlen = ps[0] + 1; tp = rb + ps[0]; qp = cs + cl - 1; st = -1; }          // src/gen_cons.c:218
                else { const int lastp = ps[T.pos_n - 1]; tlen = L - lastp - 1; tp = rb + lastp + 1; qp = cs; st = 1; }   // :221
                if (tlen > 0) {
                    if ((long long)((tlen + 63) & ~63) * 2 > bnd_stride) { if (lane == 0) atomicOr(err, 4); }
                    else {
                        // Anti-diagonal engine first, with the narrowest band of 128 NP diagonals either side whose bound (th_adx_bound) lies
                        // below the score a flank that continues the repeat is expected to reach ((100 - xpct) % of the shorter sequence).
                        // A band that cannot be certified still yields a banded maximum, a lower bound of the true one, and that names the
                        // band which is certified for sure: a wider one on the same engine, or -- beyond four registers per lane -- the
                        // strip engine with everything pruned that cannot reach that maximum (th_ksw_ext_band_wave: exact for any lower
                        // bound); pairs of a thousand bases and more whose expected score no register band reaches are probed with four
                        // registers for that purpose (an eighth of the full matrix's cost).  The full matrix is what is left.
                        bool done = false;
                        int bh = 0;
                        if (mode && cl + tlen < 30000) {
                            const int mn = min(cl, tlen), s_est = mn - (mn * xpct) / 100;
                            int np = 0;
                            for (int c = 1; c <= 4; ++c) if (th_adx_bound(c, cl, tlen) < s_est) { np = c; break; }
                            if (!np && mn >= 1024) np = xprobe;
                            while (np && cl + tlen + 512 * np + 16 <= LDSB) {
                                done = th_adx_align<4>(np, qp, st, cl, tp, st, tlen, s_seq, &mq, &mt, &bh);
#if defined(TH_ALN_PROF)
                                prof_cls = (I.which == -1 ? 5 : 7) + ((done && prof_cells == 0) ? 0 : 1);
                                prof_cells += (unsigned long long)(cl + tlen) * 128ull * (unsigned long long)np;
                                if (done) { int nn_ = 1; while (nn_ < 4 && th_adx_bound(nn_, cl, tlen) >= max(bh, 1)) ++nn_; prof_need = (unsigned long long)(max(mq, 0) + max(mt, 0) + 2) * 128ull * (unsigned long long)nn_; }
#endif
                                if (done || bh <= 0) break;
                                int np2 = 0;
                                for (int c = np + 1; c <= 4; ++c) if (th_adx_bound(c, cl, tlen) < bh) { np2 = c; break; }
                                np = np2;
                            }
                            if (!done && bh > 0) {
                                int W = 512;
                                while (W < cl + tlen && th_ext_bound_w(W, cl, tlen) >= bh) W += 64;
                                // exact for any lower bound bh (th_ksw.h); two columns per lane on packed int16 where the scores fit (round 4)
                                if (pkstrip && cl + tlen <= TH_KSW_PK_MAXLEN) th_ksw_ext_band_pk_wave(qp, st, cl, tp, st, tlen, W, bh, mybnd, s_in, &mq, &mt);
                                else th_ksw_ext_band_wave(qp, st, cl, tp, st, tlen, W, bh, mybnd, s_in, &mq, &mt);
                                done = true;
                            }
                        }
#if defined(TH_ALN_PROF)
                        prof_desc = (unsigned long long)tlen | (unsigned long long)cl << 16 | (unsigned long long)max(bh, 0) << 32 | (unsigned long long)(!done ? 3 : bh > 0 && mq >= -1 ? 2 : 1) << 60;
#endif
                        // (what no band could be staged for -- flank and consensus together longer than the staging area -- is probed on a prefix and pruned: th_aln_long_flank.
                        // Only in the longest-first instantiation, the launches with few items per wave, where one such item IS the launch's tail: the call's mere presence
                        // moved config 5's launch -- 605 items per wave, no such flank -- from 405 to 420 ms whatever its form (in line, out of line, at this site or in
                        // front of the probes, stack frame or none: five builds), and with hundreds of items per wave an outlier of 100 ms is rarely the last one running)
#if defined(TH_ALN_PROF)
                        if (prof_cls == 9) prof_cls = I.which == -1 ? 6 : 8;
#endif
                        if (!done) {
                            if (LPT && mode && longflank && cl + tlen < 30000 && th_aln_unstaged_flank<LDSB>(cl, tlen) && LDSB - cl - 16 - 1024 >= 1024) {
                                th_aln_long_flank<LDSB>(qp, st, cl, tp, tlen, s_seq, mybnd, s_in, pkstrip, &mq, &mt, &bh);
                                if (aln_stat && lane == 0) atomicAdd(&aln_stat[4], 1);       // (TH_TRACE and the debug hook: flanks probed on their prefix)
                            }
                            else th_ksw_ext_wave(qp, st, cl, tp, st, tlen, mybnd, &mq, &mt);
                        }
                        if (mode == 2) {
                            // verification: the full matrix against whatever engine produced (mq, mt), and against BOTH pruned strip engines run with the
                            // best lower bound at hand (the probe's banded maximum, or 1: any slb <= the true maximum must give the full matrix's cells)
                            int mq2 = -1, mt2 = -1;
                            th_ksw_ext_wave(qp, st, cl, tp, st, tlen, mybnd, &mq2, &mt2);
                            if ((mq2 != mq || mt2 != mt) && lane == 0) atomicOr(err, 64);
                            const int slb2 = max(bh, 1);
                            int W2 = 64;
                            while (W2 < cl + tlen && th_ext_bound_w(W2, cl, tlen) >= slb2) W2 += 64;
                            int a_ = -1, b_ = -1;
                            th_ksw_ext_band_wave(qp, st, cl, tp, st, tlen, W2, slb2, mybnd, s_in, &a_, &b_);
                            if ((a_ != mq2 || b_ != mt2) && lane == 0) atomicOr(err, 64);
                            if (cl + tlen <= TH_KSW_PK_MAXLEN) {
                                a_ = -1; b_ = -1;
                                th_ksw_ext_band_pk_wave(qp, st, cl, tp, st, tlen, W2, slb2, mybnd, s_in, &a_, &b_);
                                if ((a_ != mq2 || b_ != mt2) && lane == 0) atomicOr(err, 64);
                            }
                        }
                    }
                }
            }
            if (lane == 0) { const int o = 4 * I.task + (I.which == -1 ? 0 : 2); ext[o] = mq; ext[o + 1] = mt; }
        }
    }
    }
#if defined(TH_ALN_PROF)
    if (lane == 0 && prof_items > 0) { const unsigned long long te = __builtin_amdgcn_s_memtime() - prof_t0;
        atomicAdd(&th_aln_prof[0], 1ull); atomicMin(&th_aln_prof[1], te); atomicMax(&th_aln_prof[2], te); atomicAdd(&th_aln_prof[3], te >> 10); atomicAdd(&th_aln_prof[4], (unsigned long long)prof_items); atomicAdd(&th_aln_prof[8 + min((int)(te >> 25), 31)], 1ull); }
#endif
}
#undef xprobe
#undef longflank
#undef pkstrip
#undef xpct
