// th_k_chain.hip -- stage 2: chaining DP over self-hits and score-ranked chain extraction (gfx950).
//
// Replaces tandem_chain() and helpers, src/tandem_chain.c:290-404 (get_con_score :151-166,
// init_dp :113-126, sort_dp_score :32-43, backtrack_dp :86-111, is_in_chain :170-185,
// is_overlap_chain :54-83, sort_chain :188-207, copy_chain :209-223, set_start_period :251-255).
// With -w 1 every DP row holds exactly one cell (one hit per end position), so a "cell" is a
// hit index.
#include "th_dev.h"

#define TH_CHAIN_BLOCK 256      // th_k_chain_dp: 4 waves = 4 reads per workgroup
#define TH_TOPN 1000            // src/tandem_chain.c:361
#define TH_XSORT_LDS 8192       // candidates sorted in LDS up to this many hits, else in HBM scratch
#define TH_XBLOCK 256
#define TH_XFAST_MAX 8192       // reads with at most this many hits take the one-wave-per-read extraction kernel
#define TH_XFAST_SMALL 4608     // ... and those with at most this many its low-LDS instantiation

enum { TH_NO_CON = 0, TH_REG_CON = 1, TH_SAME_CON = 2, TH_OVL_CON = 3 };
#if defined(TH_CHAIN_PROF)
// -DTH_CHAIN_PROF=1 (tools/build_variant.sh; never the product): [0] heads scanned, [1] trips of 64 predecessor evaluations, [2] hits, [3] followers -- the chain DP's
// "cells" for the bench line's secondary roofline (pair evaluations = 64 x trips); printed and reset by th_pipeline.hip after every batch
__device__ unsigned long long th_chain_prof[8];      // [4] heads that scanned a list, [5] their trips, [6] list entries written, [7] lists
#endif

// ---------------------------------------------------------------------------------------------
// Chain DP, one read per wavefront.  `cur` advances sequentially (its predecessors' scores must be
// final); the scan over predecessors cur-1, cur-2, ... runs 64 at a time.  The reference's
// order-dependent early exits (src/tandem_chain.c:331-350) are reproduced exactly: an exclusive
// prefix-max of the candidate scores, seeded with the running best, tells every lane whether it
// would have improved the best in sequential order; ballots of the four stop rules give the
// first lane (in scan order) that ends the scan, and lanes beyond it are discarded.
// ---------------------------------------------------------------------------------------------
// PK: every read of the batch is shorter than 65536 bases, so a hit's (end, period) fit one 32-bit word; the window
// scan then reads one 8-byte {end<<16|period, score} record per predecessor instead of three 4-byte words.
// RING: the packed records of the most recent RING hits of each wave's read are staged in LDS (4 KiB per wave, 16 KiB per
// workgroup: still 8 waves per SIMD).  A record enters the ring once, 64 at a time and coalesced, when `cur` reaches its chunk;
// its score is patched in place when the hit is finalised; the window scan then reads LDS instead of re-reading every hit ~250
// times through L2 (the in-flight working set of 8192 waves x 34 KB does not fit L2, so those re-reads were fabric traffic:
// 888 GB per 100 000 reads, profiles/r01_hbm_traffic.json).  Trips that reach further back than the ring (rare: more than
// ~450 hits within one period) read HBM as before.
// ---- PERIOD-CLASS LISTS (round 6; CLS).  A head evaluates 836 predecessors on average at the headline shape and 230 of them can connect at all: a connection needs
// 5 cp < 9 pp and 5 pp < 9 cp (src/tandem_chain.c:154), and a third of a tandem repeat's hits carry two, three, ... unit lengths as period, so every 64-hit block of the
// window holds hits of every class and nothing can be skipped block-wise (docs/HISTORY.md).  Instead each read keeps index-ordered LISTS of its finished hits by period class:
// periods are binned 12 per octave (th_pbin); a bin gets a list when it holds at least TH_CLS_MIN_HEADS run heads, its heads' windows are long and the list would filter
// (below); the list holds ALL hits whose period lies in (5 lo / 9, ceil(9 hi / 5)) for the bin's periods [lo, hi] -- everything a head of the bin can connect to, ~6 % more at
// either end; the scan's own period test stays.  A list entry (ThClsEnt, 12 bytes) is the hit's packed record {end << 16 | period, FINAL score} plus its index;
// the hits of a 64-hit chunk are appended when the chunk is done (scores final, from registers), so the lists are contiguous, in index order, and a trip over a list is one
// coalesced load like a trip over the hit array.  A head of a listed bin scans (i) the hits of its own chunk before it -- one unfiltered trip from the LDS ring, as every scan
// starts -- and (ii) its list from the chunk's start downwards: the same trip, order and rules as the scan of the hit array; the hits it never sees are NO_CON ones, which
// count in the max_h rule only, and that rule is evaluated on hit INDICES.  (A first version kept 16-bit indices only and gathered the records by them: 20 % fewer
// instructions and 134 ms against 93 -- the gathers tripled the kernel's line traffic; with bins of ratio 1.26 every two-period list swallowed the one-period class: 99 ms.)  The list area holds TH_CLS_DUP entries per base
// slot of the read; lists that do not fit are dropped and their heads scan the hit array as before: same results either way (TH_CHAIN_CLS=0: no lists at all).
#define TH_CLS_NB 120           // period bins: 12 per octave (ratio 1.059) from period 64 up, periods < 65536.  Narrow on purpose: a tandem repeat's classes sit at ratio 2 (one and two
                                // unit lengths) and a connection needs ratio < 1.8, so a bin's list may reach only ~10 % beyond what its heads can connect to or it swallows the next class
#define TH_CLS_MAXL 24          // lists per read
#ifndef TH_CLS_MIN_HEADS
#define TH_CLS_MIN_HEADS 6      // a bin gets a list when at least this many heads would use it (measured at the headline shape: 32 -> 84.3 ms, 16 -> 78.6, 8 -> 75.6, 4 -> 74.8, 2 -> 76.1) ...
#endif
#ifndef TH_CLS_MIN_WINDOW
#define TH_CLS_MIN_WINDOW 512    // ... and its heads' windows hold about this many hits or more
#endif
#ifndef TH_CLS_MAX_SHARE
#define TH_CLS_MAX_SHARE 8       // ... and it holds at most this many sixteenths of the read's hits
#endif
#define TH_CLS_DUP 3            // list entries per base slot of the read (capacity of its list area: 12 bytes each)
// bin of a period: 12 (e - 6) + the number of thresholds T_i = round(1024 x 2^(i / 12)) at or below its 11-bit mantissa; -1 below 64
__device__ __forceinline__ int th_pbin(int p)
{
    const int e = 31 - TH_CLZ_NZ(p | 1), x = (int)(((uint32_t)p << 10) >> e);
    const int f = (int)(x >= 1085) + (int)(x >= 1149) + (int)(x >= 1218) + (int)(x >= 1290) + (int)(x >= 1367) + (int)(x >= 1448) + (int)(x >= 1534) + (int)(x >= 1625) + (int)(x >= 1722) + (int)(x >= 1825) + (int)(x >= 1933);
    return p < 64 ? -1 : 12 * (e - 6) + f;
}
__device__ __forceinline__ int th_pbin_lo(int b)      // smallest period of bin b (b = TH_CLS_NB: one past the largest period)
{
    const int e = b / 12 + 6, f = b - 12 * (e - 6);
    const int t = f == 0 ? 1024 : f == 1 ? 1085 : f == 2 ? 1149 : f == 3 ? 1218 : f == 4 ? 1290 : f == 5 ? 1367 : f == 6 ? 1448 : f == 7 ? 1534 : f == 8 ? 1625 : f == 9 ? 1722 : f == 10 ? 1825 : 1933;
    return (int)((((uint32_t)t << e) + 1023u) >> 10);
}

struct ThClsEnt { uint32_t ep; int sc; uint32_t ix; };      // a list entry: end << 16 | period, final score, hit index (one 12-byte load per lane and trip)
template <bool PK, int RING, bool CLS = false>      // RING: entries of the LDS ring (0 = none, 512, 768, 1024); slot of hit x = x mod RING
__global__ void __launch_bounds__(TH_CHAIN_BLOCK)
th_k_chain_dp(const uint32_t *__restrict__ roff, int n_reads, ThDevPara P,
              const int *__restrict__ hit_end, const int *__restrict__ hit_per, const int *__restrict__ hit_n,
              int *__restrict__ score, int *__restrict__ from, unsigned long long *__restrict__ eps, ThClsEnt *__restrict__ cls_ent = nullptr)
{
    static_assert(!CLS || (PK && RING >= 2 * TH_WAVE && (RING & (RING - 1)) == 0), "the class lists hold 16-bit hit indices; a listed head reads its own chunk from the ring");
    // Measured alternatives (batch 16384, config 3; this version 43 ms): a per-wave LDS ring of the last 1024 hits
    // removes the window re-reads from HBM but its 48 KiB per workgroup cuts residency to 3 waves per SIMD (62 ms);
    // a 512-hit ring 46 ms; scoring 2 or 4 chunks ahead of the resolution 49 / 85 ms.  Each hit is re-read ~250 times by
    // the sliding window (881 GB of L2/Infinity-Cache fabric traffic per 65536 reads = 5.5 TB/s), with VALU issue
    // (about 90 instructions per 64 predecessors) right behind; hence the packed 8-byte records (161 -> 151 ms per
    // 65536 reads).  Two predecessors per lane (128 per trip, shared scan and ballots) measured 154 ms: the per-predecessor
    // scoring, not the scan, is what issues.
    const int lane = TH_LANE();
    const int r = blockIdx.x * (TH_CHAIN_BLOCK / TH_WAVE) + (threadIdx.x >> 6);
    const bool live = r < n_reads;
    const uint32_t off = live ? roff[r] : 0;
    const int n = TH_UNI(live ? hit_n[r] : 0), k = P.k;
    const int *E = hit_end + off, *Pd = hit_per + off;
    int *S = score + off, *F = from + off;
    unsigned long long *R = eps + off;
    __shared__ unsigned long long s_ring[RING ? TH_CHAIN_BLOCK / TH_WAVE : 1][RING ? RING : 1];
    unsigned long long *ring = s_ring[RING ? (threadIdx.x >> 6) : 0];
    constexpr bool RP2 = RING == 0 || (RING & (RING - 1)) == 0;         // power of two: the slot is a mask; else the wave keeps cur mod RING in a scalar
    int cur_slot = 0;                                                   // cb mod RING (rings whose size is not a power of two)
    // init_dp (src/tandem_chain.c:113-126): score = k + min(k, period), no predecessor
    for (int i = lane; i < n; i += TH_WAVE) {
        const int e = E[i], p = Pd[i], s0 = k + min(k, p);
        S[i] = s0; F[i] = -1;
        if (PK) R[i] = (unsigned long long)(uint32_t)s0 << 32 | (uint32_t)e << 16 | (uint32_t)p;
    }
    TH_WAVE_FENCE();
    if (RING && lane < n) ring[lane] = R[lane];                      // chunk 0
    // end / period of the 64 hits of cur's chunk, one per lane: the current hit's pair is a v_readlane away -- a load per hit
    // (two dependent round trips to L2 before the first trip could start) was a quarter of the kernel's time
    int c_e = lane < n ? E[lane] : 0, c_p = lane < n ? Pd[lane] : 1;
    // ---- class lists of this read (see above): three passes over its hits, 64 at a time
    __shared__ int s_cls[CLS ? TH_CHAIN_BLOCK / TH_WAVE : 1][CLS ? 128 : 1];       // per wave: [bin] heads of the bin, then the list of the bin (or -1)
    ThClsEnt *lst_ent = nullptr;
    int v_lbase = 0, v_run = 0, v_llo = 0, v_lhi = 0;      // lane j: list j's first entry in the read's list area / entries appended so far / its periods: llo < pp < lhi
    int n_lists = 0;
    if (CLS && n > 2 * TH_WAVE) {
        int *s_b2l = s_cls[threadIdx.x >> 6];
        lst_ent = cls_ent + (size_t)TH_CLS_DUP * off;
        const int cap_entries = TH_UNI(TH_CLS_DUP * (int)(roff[r + 1] - off));
        s_b2l[lane] = 0; s_b2l[lane + TH_WAVE] = 0;
        TH_WAVE_FENCE();
        // pass 0: run heads per period bin
        {
            int pe_ = 0, pp_ = 0;
            for (int cb = 0; cb < n; cb += TH_WAVE) {
                const bool okl = cb + lane < n;
                const int e_ = okl ? E[cb + lane] : 0, p_ = okl ? Pd[cb + lane] : 1;
                const int qe = th_shr1(e_, pe_), qp = th_shr1(p_, pp_), d_ = e_ - qe;
                const bool fol = okl && (cb + lane) >= 1 && p_ == qp && d_ >= 1 && d_ < k && d_ <= p_;
                const int b_ = th_pbin(p_);
                if (okl && !fol && cb + lane >= 1 && b_ >= 0) atomicAdd(&s_b2l[b_], 1);
                pe_ = TH_READLANE(e_, TH_WAVE - 1); pp_ = TH_READLANE(p_, TH_WAVE - 1);
            }
        }
        TH_WAVE_FENCE();
        // the bins that get a list (lane l looks after bins l and l + 64): enough heads, and LONG windows only -- a head's window holds about cp x (hits per base) hits, and below
        // TH_CLS_MIN_WINDOW of them the scan of the hit array, its first trips served by the LDS ring, is cheaper than a list's upkeep (the main period's own class would be the
        // largest list by far: two thirds of the hits)
        const int win_thr = TH_UNI((int)(((long long)TH_CLS_MIN_WINDOW * (long long)(roff[r + 1] - off) + n - 1) / n));
        const int b0 = lane, b1 = lane + TH_WAVE;
        const bool sel0_ = s_b2l[b0] >= TH_CLS_MIN_HEADS && th_pbin_lo(b0) >= win_thr, sel1_ = b1 < TH_CLS_NB && s_b2l[b1] >= TH_CLS_MIN_HEADS && th_pbin_lo(min(b1, TH_CLS_NB - 1)) >= win_thr;
        unsigned long long selm0 = TH_BALLOT(sel0_), selm1 = TH_BALLOT(sel1_);
        while (__popcll(selm0) + __popcll(selm1) > TH_CLS_MAXL) { if (selm0) selm0 &= selm0 - 1; else selm1 &= selm1 - 1; }      // (more than 24: the shortest periods go without)
        const int n0 = (int)__popcll(selm0);
        const int j0 = (int)__popcll(selm0 & ((1ull << lane) - 1)), j1 = n0 + (int)__popcll(selm1 & ((1ull << lane) - 1));
        const bool s0 = (selm0 >> lane) & 1, s1 = (selm1 >> lane) & 1;
        n_lists = n0 + (int)__popcll(selm1);
        TH_WAVE_FENCE();
        s_b2l[b0] = s0 ? j0 : -1; s_b2l[b1] = s1 ? j1 : -1;
        // a list's periods: everything a head of the bin can connect to -- 5 cp < 9 pp and 5 pp < 9 cp for cp in [lo, hi] <=> (5 lo) / 9 < pp < ceil(9 hi / 5) -- handed from the bin's lane to lane j through LDS
        __shared__ int s_rng[CLS ? TH_CHAIN_BLOCK / TH_WAVE : 1][CLS ? 2 * TH_CLS_MAXL : 1];
        int *s_r = s_rng[threadIdx.x >> 6];
        if (s0) { s_r[2 * j0] = (5 * th_pbin_lo(b0)) / 9; s_r[2 * j0 + 1] = (9 * (th_pbin_lo(b0 + 1) - 1) + 4) / 5; }
        if (s1) { s_r[2 * j1] = (5 * th_pbin_lo(b1)) / 9; s_r[2 * j1 + 1] = (9 * (th_pbin_lo(b1 + 1) - 1) + 4) / 5; }
        TH_WAVE_FENCE();
        if (lane < n_lists) { v_llo = s_r[2 * lane]; v_lhi = s_r[2 * lane + 1]; }
        // pass 1: entries per list
        int v_cnt = 0;                     // lane j: entries of list j
        for (int cb = 0; cb < n; cb += TH_WAVE) {
            const uint32_t p_ = cb + lane < n ? (uint32_t)Pd[cb + lane] : 0u;
            for (int j = 0; j < n_lists; ++j) {
                const uint32_t lo_ = (uint32_t)TH_READLANE(v_llo, j), hi_ = (uint32_t)TH_READLANE(v_lhi, j);
                const int c_ = (int)__popcll(TH_BALLOT(p_ - lo_ - 1u < hi_ - lo_ - 1u));
                if (lane == j) v_cnt += c_;
            }
        }
        // a list pays only if it FILTERS: one that holds more than TH_CLS_MAX_SHARE / 16 of the read's hits (the main period's own class, when that is what has the long windows:
        // config 4) costs its upkeep and the dearer trip and saves nothing -- dropped, the others close ranks (lane j -> its rank among the kept) and the bins follow
        {
            const unsigned long long keepm = TH_BALLOT(lane < n_lists && 16 * v_cnt <= TH_CLS_MAX_SHARE * n);
            if (keepm != (n_lists >= 64 ? ~0ull : (1ull << n_lists) - 1)) {
                const int nj = (int)__popcll(keepm & ((1ull << lane) - 1));
                const bool kp = (keepm >> lane) & 1;
                TH_WAVE_FENCE();
                if (lane < n_lists) s_r[lane] = kp ? nj : -1;                     // old list -> new list
                TH_WAVE_FENCE();
                { const int o0 = s_b2l[b0], o1 = s_b2l[b1]; const int m0 = o0 >= 0 ? s_r[o0] : -1, m1 = o1 >= 0 ? s_r[o1] : -1; TH_WAVE_FENCE(); s_b2l[b0] = m0; s_b2l[b1] = m1; }
                TH_WAVE_FENCE();
                if (kp) { s_r[nj] = v_llo; s_r[TH_CLS_MAXL + nj] = v_lhi; }
                TH_WAVE_FENCE();
                const int nl_ = (int)__popcll(keepm);
                // counts move with their lists
                int moved = 0;
                for (int j = 0; j < n_lists; ++j) { const int c_ = TH_READLANE(v_cnt, j); const int t_ = TH_READLANE(kp ? nj : -1, j); if (t_ >= 0 && lane == t_) moved = c_; }
                v_cnt = moved;
                if (lane < nl_) { v_llo = s_r[lane]; v_lhi = s_r[TH_CLS_MAXL + lane]; } else { v_llo = 0; v_lhi = 0; }
                n_lists = TH_UNI(nl_);
                TH_WAVE_FENCE();
            }
        }
        // lists that do not fit the read's area are dropped from the top (their heads scan the hit array)
        {
            int tot = 0;
            for (int j = 0; j < n_lists; ++j) { const int c_ = TH_READLANE(v_cnt, j); if (tot + c_ > cap_entries) { n_lists = j; break; } if (lane == j) v_lbase = tot; tot += c_; }
        }
        n_lists = TH_UNI(n_lists);
#if defined(TH_CHAIN_PROF)
        if (lane == 0) atomicAdd(&th_chain_prof[7], (unsigned long long)n_lists);
#endif
        TH_WAVE_FENCE();
    }
    int c_hj = -1;                         // this lane's hit: the list a head of its bin scans (or -1)
    if (CLS && n_lists && lane < n) { const int b_ = th_pbin(c_p); const int j_ = b_ >= 0 ? s_cls[threadIdx.x >> 6][b_] : -1; c_hj = j_ < n_lists ? j_ : -1; }
    // FOLLOWERS.  A hit whose immediate predecessor has the same period and ends fewer than k bases before it -- the next k-mer of a
    // run of matching k-mers, ~80 % of the hits of a tandem repeat at 10 % error -- needs no scan: that predecessor is a connection
    // of type OVL (:151-166: same period, matched = 2 d < 2 k), an OVL connection ends the scan whether it improves or not
    // (:342-346), and it always improves, because its score is the predecessor's (>= the common initial score k + min(k, period))
    // plus matched - gap = 2 d - ilog2(2 d) / 2 >= 2.  So score = predecessor's score + that constant and from = the predecessor:
    // along a run the scores are a prefix sum over the distances, taken 64 hits at a time, and only the HEADS of runs are scanned.
    // (What the sequential scan of such a hit did: one trip of 64 lanes, of which lane 0 decided.)
    // (Tried on top of this and dropped: skipping whole trips whose 64-hit blocks cannot beat the running best -- block maxima of the final
    // scores in lane registers, + 2k as the largest connection score.  A third of a tandem repeat's hits have periods of two, three, ...
    // unit lengths (their copy one period back was hit by an error); their running best stays low while every block's maximum comes
    // from the main chain, so the bound skips 12 % of the trips, 24 % with maxima per period class (ratio 1.8), 44 % with an exact
    // per-trip filter -- and the bookkeeping per trip cost more than that: 125 -> 234 ms per 100 000 reads.)
    int prev_e = 0, prev_p = 0, carry_sc = 0;         // the hit before the chunk: end, period, final score
    for (int cb = 0; cb < n; cb += TH_WAVE) {
        if (cb) {
            if (RING) {                                                // the chunk cb .. cb+63 enters the ring (its scores are still the initial ones)
                if (cb + lane < n) ring[RP2 ? ((cb + lane) & (RING - 1)) : (int)min((uint32_t)(cur_slot + lane), (uint32_t)(cur_slot + lane - RING))] = R[cb + lane];
                TH_WAVE_FENCE();
            }
            if (cb + lane < n) { c_e = E[cb + lane]; c_p = Pd[cb + lane]; }
            if (CLS && n_lists) { c_hj = -1; if (cb + lane < n) { const int b_ = th_pbin(c_p); const int j_ = b_ >= 0 ? s_cls[threadIdx.x >> 6][b_] : -1; c_hj = j_ < n_lists ? j_ : -1; } }
        }
        const int ring_lo = RING ? cb + TH_WAVE - RING : 0;      // oldest hit index the ring still holds
        const bool okl = cb + lane < n;
        const int q_e = th_shr1(c_e, prev_e), q_p = th_shr1(c_p, prev_p);      // the hit before this lane's
        const int d = c_e - q_e;
        const bool fol = okl && (cb + lane) >= 1 && c_p == q_p && d >= 1 && d < k && d <= c_p;
        const unsigned long long follm = TH_BALLOT(fol), okm = TH_BALLOT(okl);
#if defined(TH_CHAIN_PROF)
        if (lane == 0) { atomicAdd(&th_chain_prof[2], (unsigned long long)__popcll(okm)); atomicAdd(&th_chain_prof[3], (unsigned long long)__popcll(follm)); }
#endif
        const int fcon = fol ? 2 * d - ((31 - TH_CLZ_NZ(2 * d)) >> 1) : 0;
        const int ps = th_wave_incl_sum(fcon);                  // followers: score = score of the run's head + ps - ps[head]
        int seg_sc = carry_sc, seg_ps = 0, seg_lo = 0;          // the segment before the chunk's first head continues the previous chunk's last run
        unsigned long long todo = okm & ~follm;                 // heads of the chunk, in order
        int fin = 0;                                            // final score of this lane's hit
        for (;;) {
            const int h = todo ? th_ctz64nz(todo) : TH_WAVE;    // next head (64: none left)
            // followers between the previous head and this one
            const unsigned long long segm = seg_lo >= TH_WAVE ? 0ull : follm & (h >= TH_WAVE ? ~0ull : ((1ull << h) - 1)) & (~0ull << seg_lo);
            if (segm) {
                if (TH_LANE_OF_MASK(segm, lane)) {
                    const int sc = seg_sc + ps - seg_ps, x = cb + lane;
                    fin = sc;
                    S[x] = sc; F[x] = x - 1;
                    if (PK) ((int *)(R + x))[1] = sc;
                    if (RING) ((int *)(ring + (RP2 ? (x & (RING - 1)) : (int)min((uint32_t)(cur_slot + lane), (uint32_t)(cur_slot + lane - RING)))))[1] = sc;
                }
                TH_WAVE_FENCE();
            }
            if (!todo) break;
            todo &= todo - 1;
            const int cur = cb + h;
            const int ce = TH_READLANE(c_e, h), cp = TH_READLANE(c_p, h), cs = ce - cp;       // the same in every lane: scalar registers
            const int init = k + min(k, cp);
            int best = init + 15, bi = -1, iter = 0;       // best and the candidates are kept 15 above their value inside the scan (the constant of the logarithm term)
            if (cur >= 1) {
        const int max_h = cp;
        // the two period-ratio tests below as one range test on the predecessor's period: 5 cp >= 9 pp <=> pp <= (5 cp) / 9 and
        // 5 pp >= 9 cp <=> pp >= ceil(9 cp / 5), so a connection needs plo < pp < phi
        const uint32_t plo = (5u * (uint32_t)cp) / 9u, phi = (9u * (uint32_t)cp + 4u) / 5u;
        const int hslot = RP2 ? 0 : (cur_slot + h >= RING ? cur_slot + h - RING : cur_slot + h);        // cur mod RING
        int bslot = RP2 ? 0 : (hslot == 0 ? RING - 1 : hslot - 1);      // (cur - 1) mod RING
        const int hj = (CLS && n_lists) ? TH_READLANE(c_hj, h) : -1;
        if (CLS && hj >= 0) {
            // ---- the head's bin has a list (see the top of the kernel).  One trip: 64 predecessors, lane l holding hit `idx`; `livem`: the lanes that hold one.  The rules of
            // the scan below, with the max_h rule (:349) on hit indices: it counts the hits scanned since the last improvement, listed or not
            int last_imp = cur;                         // hit index of the last improvement
            auto cls_trip = [&](const uint32_t ep, const int psc, const int idx, const unsigned long long livem) -> unsigned long long {
                const int pe = (int)(ep >> 16), pp = (int)(ep & 0xffffu);
                const int ps = pe - pp;
                const unsigned long long dstopm = livem & TH_BALLOT(pe < cs);                              // :331
                const unsigned long long conm = livem & ~dstopm & TH_BALLOT(cs > ps) & TH_BALLOT((uint32_t)pp - plo - 1u < phi - plo - 1u);
                const int dis1 = ce - pe, dis2 = cs - ps;
                const int matched = min(dis1, k) + min(dis2, k);
                const int dq = cp - pp;
                const int dq2h = (int)((uint32_t)__mul24(dq, dq) >> 1);
                const int clzh = TH_CLZ_NZ(dis1 + dis2) >> 1;
                const int s = TH_LANE_OF_MASK(conm, lane) ? (psc + matched + clzh) - dq2h : -0x40000000;
                unsigned long long impm = TH_BALLOT(s > best);
                const unsigned long long samem = conm & TH_BALLOT(cp == pp);
                unsigned long long stopm = dstopm | (samem & TH_BALLOT(matched < 2 * k));
                // the max_h rule can fire in this trip only if its lowest index lies max_h hits or more below the last improvement (one inside the trip only moves that further down)
                const bool mh = last_imp - TH_READLANE(idx, th_msb64nz(livem)) >= max_h;
                if (impm) {
                    if (impm & (impm - 1)) {
                        const int rb = max(th_shr1_z(th_wave_incl_max(s)), best);
                        impm = TH_BALLOT(s > rb);
                    }
                    if (mh) {
                        // per lane the hit index of the last improvement before it in scan order; a hit MORE than max_h below that was never reached by the scan of the hit
                        // array (it had stopped at an unlisted hit): whatever it scores, the scan ends in front of it
                        const unsigned long long below = impm & ((1ull << lane) - 1);
                        const int sh_ = __shfl(idx, th_msb64(below | 1ull));
                        const int li = below ? sh_ : last_imp;
                        const unsigned long long deadm = livem & TH_BALLOT(li - idx > max_h);
                        impm &= ~deadm;
                        stopm |= deadm | (livem & ~impm & TH_BALLOT(li - idx >= max_h));
                    }
                    stopm |= samem & impm;
                    const unsigned long long taken = impm & (stopm ^ (stopm - 1));
                    if (taken) { const int last = th_msb64nz(taken); best = TH_READLANE(s, last); bi = TH_READLANE(idx, last); last_imp = bi; }
                } else if (mh) stopm |= livem & TH_BALLOT(last_imp - idx >= max_h);
                return stopm;
            };
#if defined(TH_CHAIN_PROF)
            int prof_trips = 0;
#endif
            unsigned long long ends = 0ull;
            // (i) the hits of this chunk before the head, all of them: hit cur - 1 - lane for lane < h, from the LDS ring (where their scores were patched as they finished)
            if (h > 0) {
                const int idx = cur - 1 - lane;
                const unsigned long long rec = ring[idx & (RING - 1)];
                ends = cls_trip((uint32_t)rec, (int)(rec >> 32), idx, (1ull << h) - 1);
#if defined(TH_CHAIN_PROF)
                ++prof_trips;
#endif
            }
            ends = ((unsigned long long)(uint32_t)TH_UNI((int)(ends >> 32)) << 32) | (uint32_t)TH_UNI((int)ends);
            // (ii) the list, from the entries appended before this chunk downwards: contiguous records and indices, the next trip's requested a trip ahead
            int pos = TH_READLANE(v_run, hj) - 1;
            if (!ends && pos >= 0) {
                const ThClsEnt *ents = lst_ent + TH_READLANE(v_lbase, hj);
                int li = pos - lane;                                   // this lane's entry of the trip at hand (negative: none)
                ThClsEnt ea = ents[max(li, 0)], eb;
                unsigned long long stop_ = 0ull;
                // full trips (every lane holds an entry: the masks of the rules need no "lanes that hold one"), two per round of the loop -- the entries of a trip are requested
                // during the trip before, into the OTHER register set: a copy of a loaded register at the loop's bottom would wait for the load right there -- then the
                // list's last, partial trip
#define TH_CLS_FULL_TRIP(CUR, NXT) { li -= TH_WAVE; NXT = ents[max(li, 0)]; stop_ = cls_trip(CUR.ep, CUR.sc, (int)CUR.ix, ~0ull); \
                                     stop_ = ((unsigned long long)(uint32_t)TH_UNI((int)(stop_ >> 32)) << 32) | (uint32_t)TH_UNI((int)stop_); pos -= TH_WAVE; }
                while (pos >= TH_WAVE - 1) {
#if defined(TH_CHAIN_PROF)
                    ++prof_trips;
#endif
                    TH_CLS_FULL_TRIP(ea, eb);
                    if (stop_) break;
                    if (pos < TH_WAVE - 1) { ea = eb; break; }
#if defined(TH_CHAIN_PROF)
                    ++prof_trips;
#endif
                    TH_CLS_FULL_TRIP(eb, ea);
                    if (stop_) break;
                }
#undef TH_CLS_FULL_TRIP
                if (!stop_ && pos >= 0) {
#if defined(TH_CHAIN_PROF)
                    ++prof_trips;
#endif
                    (void)cls_trip(ea.ep, ea.sc, (int)ea.ix, (2ull << pos) - 1);
                }
            }
#if defined(TH_CHAIN_PROF)
            if (lane == 0) { atomicAdd(&th_chain_prof[0], 1ull); atomicAdd(&th_chain_prof[1], (unsigned long long)prof_trips); atomicAdd(&th_chain_prof[4], 1ull); atomicAdd(&th_chain_prof[5], (unsigned long long)prof_trips); }
#endif
        } else {
        // The packed records of a trip: 64 hits base, base-1, ... -- from the ring when the whole trip lies inside it, else from HBM (two
        // different loads under a wave-uniform branch: a select between the two POINTERS would become one flat load, which waits for the
        // wave's outstanding global stores -- the previous hit's results -- before every trip).  Lanes before the first hit read some
        // record and are discarded by the `valid` mask.  Slot of idx = base - lane: (cur_slot - (cur - base) - lane) mod RING, the
        // wave-uniform part reduced on the scalar unit.
        auto fetch_ring = [&](int base_, int bslot_) -> unsigned long long {
            const int idx_ = base_ - lane;
            return TH_LDS_READ64_ASYNC(&s_ring[RING ? (threadIdx.x >> 6) : 0][RP2 ? (idx_ & (RING - 1)) : (int)min((uint32_t)(bslot_ - lane), (uint32_t)(bslot_ - lane + RING))]);
        };
        auto fetch_hbm = [&](int base_) -> unsigned long long { return R[max(base_ - lane, 0)]; };
        // The next trip's records are requested before this trip is resolved (seven scans in eight go on): the trip is a chain of
        // dependent steps -- load, score, prefix maximum, masks, branch -- and the wave has nothing else to overlap the load with.
        const int base0 = cur - 1;
        // a trip at `base` asks for the records base - 64 .. base - 127: from the ring if base >= ring_from, from HBM if base >= 64, not at all below that
        const int ring_from = RING ? th_smax(ring_lo + 2 * TH_WAVE - 1, TH_WAVE) : 0x7fffffff;
        unsigned long long rec_next = 0ull;
        if (PK) { if (RING && base0 - (TH_WAVE - 1) >= ring_lo) rec_next = fetch_ring(base0, bslot); else rec_next = fetch_hbm(base0); }
        // (round 5: ONE exit, at the bottom -- the kernel is bound by scalar issue, and a `break` in the middle of a `for` with an increment of its own cost a dozen scalar
        // instructions of flags and re-tests per trip; base0 >= 0 here, so the first trip always runs.  The wait for the requested records is the FIRST thing of a trip:
        // at the bottom and behind the loop, hipcc merged the two waits' blocks and re-tested the exit flag after them)
#if defined(TH_CHAIN_PROF)
        int prof_trips = 0;
#endif
        int base = base0;
        unsigned long long endw = 0ull;                 // set by the trip that reaches hit 0: the loop's exit test is ONE 64-bit word, stopm | endw
        // byte offset, in the wave's ring, of this lane's record of the trip last requested (power-of-two rings): the next trip's is 64 slots below it, modulo the ring --
        // a subtraction and a mask where slot = (base - 64 - lane) & (RING - 1), times 8, plus the ring's address were four instructions a trip
        int rbo = (RING && RP2) ? ((base0 - lane) & (RING - 1)) << 3 : 0;
        for (;;) {
#if defined(TH_CHAIN_PROF)
            ++prof_trips;
#endif
            unsigned long long rec;
            TH_LDS_ASYNC_TAKE(rec, rec_next);  // (wait and copy in one asm block, before anything else reads the value: the compiler does not track the asynchronous read)
            const int idx = base - lane;
            const bool valid = idx >= 0;
            int pe, pp, psc;
            if (PK) {
                if (base >= ring_from) {
                    if (RP2) { rbo = (rbo - 8 * TH_WAVE) & (8 * RING - 1); rec_next = TH_LDS_READ64_ASYNC((const unsigned long long *)((const char *)ring + rbo)); }
                    else rec_next = fetch_ring(base - TH_WAVE, bslot >= TH_WAVE ? bslot - TH_WAVE : bslot - TH_WAVE + RING);
                } else if (base >= TH_WAVE) rec_next = fetch_hbm(base - TH_WAVE);
                pe = (int)((uint32_t)rec >> 16); pp = (int)((uint32_t)rec & 0xffffu); psc = (int)(rec >> 32);
            } else { pe = valid ? E[idx] : 0; pp = valid ? Pd[idx] : 1; psc = valid ? S[idx] : 0; }
            const int ps = pe - pp;
            // The rules are evaluated as 64-bit lane masks of single comparisons, combined on the scalar unit (a boolean expression per
            // lane would be materialised as 0/1 registers and compared again for every ballot).
            unsigned long long dstopm = TH_BALLOT(pe < cs);                              // :331  predecessor ends before cur starts
            // ... or does not exist (idx = base - lane < 0): the last trip only, under a real branch (the volatile asm keeps hipcc from turning it into shifts and
            // selects on every trip); base == 63 has all its lanes and only ends the loop
            if (base < TH_WAVE) { unsigned long long none = (~0ull << base) << 1; TH_OPAQUE_S(none); dstopm |= none; endw = 1ull; }
            // :154 compares cp >= pp * 1.8 in double.  fl(1.8) exceeds 1.8 by a relative 2.5e-17, less than half an ulp,
            // so pp * fl(1.8) rounds to exactly 1.8 pp whenever that is an integer and stays > 0.19 away from every
            // integer otherwise: the comparison equals 5 cp >= 9 pp for all periods below 2^28 (th_gpu_upload checks).
            const unsigned long long conm = ~dstopm & TH_BALLOT(cs > ps) & TH_BALLOT((uint32_t)pp - plo - 1u < phi - plo - 1u);     // anything but NO_CON
            // Every lane scores (a skipped lane costs the same issue slot as a computing one); lanes without a connection
            // produce garbage that the select below discards.  Hits are sorted by end, so pe <= ce, and cs > ps holds where it
            // matters: both distances are plain differences and their sum is positive there (ilog2 needs no guard).
            const int dis1 = ce - pe, dis2 = cs - ps;
            const int matched = min(dis1, k) + min(dis2, k);
            const int dq = cp - pp;
            // :128-142  gap = dq^2 / 2 + ilog2(dis1 + dis2) / 2.  Packed records mean periods below 65536: the square is a full-rate
            // 24-bit multiply (v_mul_lo_u32 runs at quarter rate), and (31 - clz(x)) >> 1 == 15 - (clz(x) >> 1) folds into the sums
            const int dq2h = (int)((uint32_t)(PK ? __mul24(dq, dq) : dq * dq) >> 1);
            const int clzh = TH_CLZ_NZ(dis1 + dis2) >> 1;
            const int s = TH_LANE_OF_MASK(conm, lane) ? (psc + matched + clzh) - dq2h : -0x40000000;        // score + 15
            // Which lanes improve the best IN SCAN ORDER.  Round 5: a lane can only improve if it beats the running best the trip started with, and when at most ONE lane
            // does, that lane is the improvement (everything in front of it is <= best) -- no prefix maximum needed: one compare and a population count instead of the
            // nine vector instructions of the scan (six dependent DPP steps, their copy, the shift, the maximum) on all but the trips with several candidates
            // (a scan's first trip or two: afterwards the best is high and candidates are rare).  15 x 10^9 of the kernel's 67 x 10^9 vector instructions.
            unsigned long long impm = TH_BALLOT(s > best);     // (a lane without a connection holds -2^30, below every best)
            // stop rules in scan order: :331 (dstop); :342 improving SAME/OVL and :344-346 non-improving OVL (SAME with fewer than 2k
            // matched bases), so OVL always stops; :349 max_h consecutive non-improving predecessors (only evaluated when reachable).
            // Most trips have NO candidate (the best is high after a scan's first trips): they pay one 64-bit test for everything a candidate entails -- the prefix
            // maximum, the improving-SAME stops, the improvements taken -- and the max_h rule in its candidate-free form
            const unsigned long long samem = conm & TH_BALLOT(cp == pp);
            unsigned long long stopm = dstopm | (samem & TH_BALLOT(matched < 2 * k));
            if (impm) {
                if (impm & (impm - 1)) {
                    const int rb = max(th_shr1_z(th_wave_incl_max(s)), best);             // best as seen just before this lane (lane 0: the running best; zero shifted into lane 0 -- one instruction less than minus infinity -- and best >= 16)
                    impm = TH_BALLOT(s > rb);
                }
                stopm |= samem & impm;
                if (iter + TH_WAVE >= max_h) {
                    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1);
                    const unsigned long long below = impm & upto;
                    const int iter_after = below ? lane - th_msb64(below) : iter + lane + 1;
                    stopm |= ~impm & TH_BALLOT(iter_after >= max_h);
                }
                // the improvements taken are those up to the first stopping lane; the last of them is the new best.  ONE update for both outcomes of the trip
                // (round 5): stopm ^ (stopm - 1) is every lane up to and including the lowest stopping one and ALL lanes when none stops, so the scan that goes on
                // and the scan that ends take their improvements through the same code; `iter` of a scan that ends is never read
                const unsigned long long taken = impm & (stopm ^ (stopm - 1));
                iter += TH_WAVE;
                if (taken) { const int last = th_msb64nz(taken); best = TH_READLANE(s, last); bi = base - last; iter = 63 - last; }
            } else {
                if (iter + TH_WAVE >= max_h) stopm |= TH_BALLOT(iter + lane + 1 >= max_h);
                iter += TH_WAVE;
            }
            // the scan ended, or this was the trip that reached hit 0: as ONE 64-bit word and one test -- two booleans become two lane masks, an AND with exec and
            // a vector-condition branch
            unsigned long long ends = stopm | endw;
            TH_OPAQUE_S(ends);
            if (ends) break;
            base -= TH_WAVE;
            if (!RP2) bslot = bslot >= TH_WAVE ? bslot - TH_WAVE : bslot - TH_WAVE + RING;
        }
#if defined(TH_CHAIN_PROF)
        if (lane == 0) { atomicAdd(&th_chain_prof[0], 1ull); atomicAdd(&th_chain_prof[1], (unsigned long long)prof_trips); }
#endif
        TH_LDS_ASYNC_WAIT(rec_next);           // a scan that stopped leaves its last request in flight: its registers stay reserved until it lands
        }
        best -= 15;
        if (lane == 0 && best > init) {                                                                 // :352-354
            S[cur] = best; F[cur] = bi;
            if (PK) ((int *)(R + cur))[1] = best;
            if (RING) ((int *)(ring + (RP2 ? (cur & (RING - 1)) : hslot)))[1] = best;
        }
        TH_WAVE_FENCE();
            } else best -= 15;
            seg_sc = max(best, init); seg_ps = TH_READLANE(ps, h); seg_lo = h + 1;
            if (lane == h) fin = seg_sc;
        }
        // the chunk's hits, finished, join the lists of their period class (records from registers: end, period, final score)
        if (CLS && n_lists) {
            const uint32_t p_ = okl ? (uint32_t)c_p : 0u;
            for (int j = 0; j < n_lists; ++j) {
                const uint32_t lo_ = (uint32_t)TH_READLANE(v_llo, j), hi_ = (uint32_t)TH_READLANE(v_lhi, j);
                const bool mem_ = p_ - lo_ - 1u < hi_ - lo_ - 1u;
                const unsigned long long mm = TH_BALLOT(mem_);
                if (mm) {
                    const int at = TH_READLANE(v_lbase, j) + TH_READLANE(v_run, j) + th_mbcnt(mm);
                    if (mem_) { ThClsEnt e_; e_.ep = (uint32_t)c_e << 16 | (uint32_t)c_p; e_.sc = fin; e_.ix = (uint32_t)(cb + lane); lst_ent[at] = e_; }
                    if (lane == j) v_run += (int)__popcll(mm);
#if defined(TH_CHAIN_PROF)
                    if (lane == 0) atomicAdd(&th_chain_prof[6], (unsigned long long)__popcll(mm));
#endif
                }
            }
            TH_WAVE_FENCE();
        }
        // the chunk's last hit is what the next chunk's first one looks back at
        const int last = min(TH_WAVE, n - cb) - 1;
        prev_e = TH_READLANE(c_e, last); prev_p = TH_READLANE(c_p, last); carry_sc = TH_READLANE(fin, last);
        if (!RP2) { cur_slot += TH_WAVE; if (cur_slot >= RING) cur_slot -= RING; }
    }
}

// ---------------------------------------------------------------------------------------------
// Chain extraction, one workgroup per read: candidates (score<<32 | hit index) are bitonic-sorted
// descending -- equal scores keep the larger hit index first, which is the order glibc's stable
// qsort leaves -- then thread 0 walks them exactly as the reference does.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TH_XBLOCK)
th_k_chain_extract(const uint32_t *__restrict__ roff, ThDevPara P,
                   const int *__restrict__ hit_end, const int *__restrict__ hit_per, const int *__restrict__ hit_n,
                   const int *__restrict__ score, const int *__restrict__ from,
                   uint8_t *__restrict__ tracked, int *__restrict__ cells,
                   unsigned long long *__restrict__ gsort, const long long *__restrict__ gsort_off,
                   ThChainRec *__restrict__ rec, int rec_cap, int *__restrict__ rec_count,
                   int *__restrict__ read_ch_base, int *__restrict__ read_ch_n, int *__restrict__ err)
{
    __shared__ unsigned long long lds[TH_XSORT_LDS];
    __shared__ int ch_off[TH_TOPN], ch_len[TH_TOPN], ch_score[TH_TOPN], ch_fs[TH_TOPN], ch_ls[TH_TOPN], ch_le[TH_TOPN];
    __shared__ short ch_idx[TH_TOPN];
    const int tid = threadIdx.x, r = blockIdx.x;
    const uint32_t off = roff[r];
    const int n = hit_n[r];
    if (n <= TH_XFAST_MAX) return;            // handled by th_k_chain_extract_wave
    const int *E = hit_end + off, *Pd = hit_per + off, *S = score + off, *F = from + off;
    uint8_t *T = tracked + off; int *CL = cells + off;
    int N = 1; while (N < n) N <<= 1;
    unsigned long long *buf = N <= TH_XSORT_LDS ? lds : gsort + gsort_off[r];
    for (int i = tid; i < N; i += TH_XBLOCK) buf[i] = i < n ? ((unsigned long long)(uint32_t)S[i] << 32 | (uint32_t)i) : 0ull;
    for (int i = tid; i < n; i += TH_XBLOCK) T[i] = 0;
    __syncthreads();
    for (int size = 2; size <= N; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (N >> 1); t += TH_XBLOCK) {
                int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                bool desc = (lo & size) == 0;
                unsigned long long a = buf[lo], c = buf[hi];
                if ((a < c) == desc) { buf[lo] = c; buf[hi] = a; }
            }
            __syncthreads();
        }
    if (tid != 0) return;

    int ch_n = 0, top = 0;
    for (int i = 0; i < TH_TOPN; ++i) ch_idx[i] = (short)i;
    for (int ci = 0; ci < n && ch_n < TH_TOPN; ++ci) {
        const int cand = (int)(uint32_t)buf[ci];
        const int cell_end = E[cand], cell_start = cell_end - Pd[cand];
        // is_in_chain, src/tandem_chain.c:170-185 (accepted chains scanned in chain_idx order)
        bool in = false;
        for (int q = 0; q < ch_n; ++q) {
            const int c = ch_idx[q];
            if (ch_len[c] <= 0) continue;
            const int cst = ch_fs[c], cen = ch_le[c];
            if (cen < cell_start) break;
            if (cst > cell_end) continue;
            if (cell_end - cst >= (cen - cst) / 2) { in = true; break; }
        }
        if (in) continue;
        // backtrack_dp, src/tandem_chain.c:86-111, into slot ch_n
        int ok = 0;
        if (!T[cand]) {
            int sc = S[cand], cur = cand, len = 0;
            for (;;) {
                T[cur] = 1; CL[top + len++] = cur;
                const int pre = F[cur];
                if (pre == -1) break;
                if (T[pre]) { sc -= S[pre]; break; }
                cur = pre;
            }
            for (int a = 0, b = len - 1; a < b; ++a, --b) { int x = CL[top + a]; CL[top + a] = CL[top + b]; CL[top + b] = x; }
            const int nw = ch_n;
            ch_off[nw] = top; ch_len[nw] = len; ch_score[nw] = sc;
            const int c0 = CL[top], c1 = CL[top + len - 1];
            ch_fs[nw] = E[c0] - Pd[c0]; ch_ls[nw] = E[c1] - Pd[c1]; ch_le[nw] = E[c1];
            if (len > 1) {
                // is_overlap_chain, src/tandem_chain.c:68-83 (and is_overlap_chain1 :54-65)
                int loses = 0;
                const int start = ch_ls[nw];
                for (int i = ch_n - 1; i >= 0; --i) {
                    if (ch_len[i] <= 0) continue;
                    if (ch_le[i] <= start) break;
                    const int s1 = ch_fs[i], e1 = ch_ls[i], s2 = ch_fs[nw], e2 = ch_ls[nw];
                    const int mn = min(e1 - s1, e2 - s2), ov = min(e1, e2) - max(s1, s2);
                    if (ov / (mn + 0.0) >= 0.5) {
                        if (ch_score[i] > ch_score[nw]) loses = 1; else ch_len[i] = 0;
                        break;
                    }
                }
                ok = !loses;
            }
        }
        if (ok) { top += ch_len[ch_n]; ++ch_n; }
        // sort_chain, src/tandem_chain.c:188-207 (selection pass keeps the reference's stale `i`)
        for (int a = 0; a + 1 < ch_n; ++a) {
            const int i = ch_idx[a];
            if (ch_len[i] <= 0) continue;
            int e1 = ch_le[i];
            for (int b = a + 1; b < ch_n; ++b) {
                const int j = ch_idx[b];
                if (ch_len[j] <= 0) continue;
                const int e2 = ch_le[j];
                if (e1 < e2) { ch_idx[a] = (short)j; ch_idx[b] = (short)i; e1 = e2; }
            }
        }
    }
    // copy_chain in reverse rank order (ascending end), chains with >= 3 cells; seed = first cell (:251-255)
    int cnt = 0;
    for (int q = ch_n - 1; q >= 0; --q) if (ch_len[ch_idx[q]] >= 3) ++cnt;
    int base = cnt ? atomicAdd(rec_count, cnt) : 0;
    if (base + cnt > rec_cap) { atomicOr(err, 1); cnt = 0; }
    read_ch_base[r] = base; read_ch_n[r] = cnt;
    if (cnt) {
        int o = base;
        for (int q = ch_n - 1; q >= 0; --q) {
            const int c = ch_idx[q];
            if (ch_len[c] < 3) continue;
            const int c0 = CL[ch_off[c]];
            ThChainRec x;
            x.read = r; x.cell_off = (int)off + ch_off[c]; x.len = ch_len[c];
            x.est_start = ch_fs[c]; x.est_period = E[c0] - ch_fs[c];
            x.first_end = E[c0]; x.last_start = ch_ls[c]; x.pad = 0;
            rec[o++] = x;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Chain extraction, one read per wavefront (reads with <= TH_XFAST_MAX hits; the kernel above is the
// fallback for longer ones).  Same decisions as the reference, organised for the wave:
//   * candidates sorted by (score desc, hit index desc) as 32-bit keys in LDS;
//   * the scan over candidates is 64-wide: a lane is skipped if its cell is covered by an accepted
//     chain (is_in_chain, :170-185) or already tracked (backtrack_dp returns at once, :87, and the
//     sort_chain that follows is a no-op on an unchanged list), the first surviving lane is attempted;
//   * backtracking follows from[] through an LDS copy, tracked flags are an LDS bitmap.
// ---------------------------------------------------------------------------------------------
// Candidate order for the one-wave extraction: (score << 13 | hit index) sorted descending by a whole workgroup
// (the sort is the only data-parallel part of the extraction); the sorted hit indices and their (start, end)
// coordinates go to HBM in candidate order so that the extraction wave reads its 64-candidate windows coalesced
// and keeps only from[] and the tracked bitmap in LDS (4 waves per CU instead of 2).
#define TH_XS_BLOCK 256
__global__ void __launch_bounds__(TH_XS_BLOCK)
th_k_chain_sort(const uint32_t *__restrict__ roff, ThDevPara P,
                const int *__restrict__ hit_end, const int *__restrict__ hit_per, const int *__restrict__ hit_n,
                const int *__restrict__ score, uint16_t *__restrict__ xorder, int *__restrict__ xstart, int *__restrict__ xend)
{
    __shared__ uint32_t s_key[TH_XFAST_MAX];
    __shared__ uint32_t s_hist[TH_XS_BLOCK / TH_WAVE][TH_RDX_BINS + 1];
    __shared__ int s_tot[16];
    const int tid = threadIdx.x, r = blockIdx.x;
    const uint32_t off = roff[r];
    const int n = hit_n[r];
    if (n > TH_XFAST_MAX || n < 2) return;
    const int *E = hit_end + off, *Pd = hit_per + off, *S = score + off;
    int N = 1; while (N < n) N <<= 1;
    if (N >= TH_XS_BLOCK && N <= TH_RDX_STEPS * TH_XS_BLOCK && !P.sort_bitonic) {
        // round 4: the same order from a stable radix sort (th_dev.h).  Slot x holds hit n - 1 - x with the COMPLEMENT of its score in the key
        // field: ascending by that field is best score first, and stability keeps slot order among equal scores = the larger index first.
        // (A score is at least k + 1, so no word equals the all-ones sentinel of the empty slots, which sink to the end.)
        for (int x = tid; x < N; x += TH_XS_BLOCK) { const int i = n - 1 - x; s_key[x] = x < n ? ((~(uint32_t)S[i] & 0x7ffffu) << 13 | (uint32_t)i) : 0xffffffffu; }
        __syncthreads();
        th_radix_sort_field<TH_XS_BLOCK>(s_key, N, 13, 19, 0xffffffffu, s_hist, s_tot, tid);
    } else {
        for (int i = tid; i < N; i += TH_XS_BLOCK) s_key[i] = i < n ? ((uint32_t)S[i] << 13 | (uint32_t)i) : 0u;
        __syncthreads();
        th_bitonic_sort<uint32_t, true, TH_XS_BLOCK>(s_key, N, tid);      // descending: best score first, ties keep the larger index first
    }
    for (int x = tid; x < n; x += TH_XS_BLOCK) {
        const int cand = (int)(s_key[x] & 0x1fff);
        const int e = E[cand];
        xorder[off + x] = (uint16_t)cand; xstart[off + x] = e - Pd[cand]; xend[off + x] = e;
    }
}

#define TH_XW_BLOCK 64
#ifndef TH_TOPL
#define TH_TOPL 96             // chain-table slots kept in LDS by the one-wave extraction: 19.5 KB per wave, 8 waves per CU (the CPU test build uses 2)
#endif
template <class T> struct ThTier {
    T *lds; int *ovf;
    __device__ __forceinline__ int get(int i) const { return i < TH_TOPL ? (int)lds[i] : TH_LOAD_L2(ovf + (i - TH_TOPL)); }
    __device__ __forceinline__ void set(int i, int v) const { if (i < TH_TOPL) lds[i] = (T)v; else ovf[i - TH_TOPL] = v; }
};
// FMAX / FMIN: the kernel takes the reads with FMIN < hits <= FMAX (and, with FMIN = 0, the reads with fewer than two hits).  Two
// instantiations share a batch: <TH_XFAST_SMALL, 0> keeps 9 KB of predecessor words per wave (14 waves per CU: the kernel is bound by
// the latency of its sequential walks, so resident waves are what it needs) and serves every read of the headline shape (4230 hits on
// average, 4521 at most), <TH_XFAST_MAX, TH_XFAST_SMALL> the reads beyond (8 waves per CU, as round 3 ran all of them).
template <int FMAX, int FMIN>
__global__ void __launch_bounds__(TH_XW_BLOCK)
th_k_chain_extract_wave(const uint32_t *__restrict__ roff, int n_reads, ThDevPara P,
                        const int *__restrict__ hit_end, const int *__restrict__ hit_per, const int *__restrict__ hit_n,
                        const int *__restrict__ score, const int *__restrict__ from, int *__restrict__ cells,
                        const uint16_t *__restrict__ xorder, const int *__restrict__ xstart, const int *__restrict__ xend, int *__restrict__ xovf,
                        ThChainRec *__restrict__ rec, int rec_cap, int *__restrict__ rec_count,
                        int *__restrict__ read_ch_base, int *__restrict__ read_ch_n, int *__restrict__ err, int *__restrict__ queue)
{
    // from[] and the tracked flags in ONE 16-bit word per hit: bits 0..13 = predecessor + 1 (0: none), bit 15 = tracked.  The backtrack's test
    // of the predecessor then reads the very word its next step needs (one dependent LDS read per cell instead of two: the walk is what
    // this kernel waits for), and a window's tracked test comes with the candidates' predecessors (round 4; 34.8 -> ms in profiles/r04_*)
    __shared__ unsigned short s_ft[FMAX];
    // chain table: the first TH_TOPL slots in LDS, the rest (up to the reference's cap of 1000 chains) in the wave's HBM
    // overflow segment -- a read rarely keeps more than a few dozen chains, and 22 KB of LDS per wave for the full table
    // held the kernel at 4 waves per CU (157 -> 84 -> 52 ms per 100 000 reads with the sort split off and 8 waves) (it is bound by the latency of its sequential attempts)
    __shared__ int l_score[TH_TOPL], l_fs[TH_TOPL], l_ls[TH_TOPL], l_le[TH_TOPL];
    __shared__ unsigned short l_off[TH_TOPL], l_len[TH_TOPL];      // offsets / lengths in the read's cell list (<= 8192 hits)
    __shared__ short l_idx[TH_TOPL];
    int *ovf = xovf + (long long)blockIdx.x * (7 * (TH_TOPN - TH_TOPL));
    const ThTier<int> ch_score = { l_score, ovf }, ch_fs = { l_fs, ovf + (TH_TOPN - TH_TOPL) }, ch_ls = { l_ls, ovf + 2 * (TH_TOPN - TH_TOPL) },
                      ch_le = { l_le, ovf + 3 * (TH_TOPN - TH_TOPL) };
    const ThTier<unsigned short> ch_off = { l_off, ovf + 4 * (TH_TOPN - TH_TOPL) }, ch_len = { l_len, ovf + 5 * (TH_TOPN - TH_TOPL) };
    const ThTier<short> ch_idx = { l_idx, ovf + 6 * (TH_TOPN - TH_TOPL) };
    const int lane = TH_LANE();
    for (;;) {
        const int r = th_queue_pop(queue);            // reads differ in hits and chains: waves pull the next read
        if (r >= n_reads) break;
        const uint32_t off = roff[r];
        const int n = hit_n[r];
        if (n > FMAX || (FMIN > 0 && n <= FMIN)) continue;
        if (n < 2) { if (lane == 0) { read_ch_base[r] = 0; read_ch_n[r] = 0; } continue; }     // src/tandem_chain.c:291
        const int *E = hit_end + off, *Pd = hit_per + off, *S = score + off, *F = from + off;
        int *CL = cells + off;
        const uint16_t *XO = xorder + off; const int *XS = xstart + off, *XE = xend + off;
        for (int i = lane; i < n; i += TH_WAVE) s_ft[i] = (unsigned short)(F[i] + 1);
        for (int i = lane; i < TH_TOPN; i += TH_WAVE) ch_idx.set(i, i);
        TH_WAVE_FENCE();
        int ch_n = 0, top = 0, ci = 0;
        while (ci < n && ch_n < TH_TOPN) {
            // ---- 64 candidates at once: covered by an accepted chain, or already tracked? ----
            const int x = ci + lane;
            const bool valid = x < n;
            const int cand = valid ? (int)XO[x] : 0;
            const int cell_end = valid ? XE[x] : 0, cell_start = valid ? XS[x] : 0;
            const int w0 = valid ? (int)s_ft[cand] : 0x8000;
            bool skip = !valid || (w0 >> 15);
            bool open = !skip;                       // is_in_chain's scan is still running for this lane
            for (int q = 0; q < ch_n; ++q) {
                const int c = ch_idx.get(q);
                if (ch_len.get(c) <= 0) continue;
                const int cst = ch_fs.get(c), cen = ch_le.get(c);
                if (open) {
                    if (cen < cell_start) open = false;
                    else if (cst > cell_end) {}
                    else if (cell_end - cst >= (cen - cst) / 2) { skip = true; open = false; }
                }
                if (!__any(open)) break;
            }
            // Candidates whose backtrack stops at once -- no predecessor, or a predecessor that is already tracked
            // (possibly by an earlier candidate of this very window) -- only get themselves tracked: the chain
            // has one cell, is rejected (:107) and the sort_chain that follows changes nothing.  They are
            // committed together up to the first candidate that needs a real walk.
            const bool nonskip = valid && !skip;
            const int pre0 = nonskip ? (w0 & 0x3fff) - 1 : -1;
            bool trivial = nonskip && (pre0 == -1 || (s_ft[pre0] >> 15));
            const unsigned long long nsm = __ballot(nonskip);
            if (nsm & (nsm - 1)) {                    // two or more live candidates: one may track another's predecessor
                for (unsigned long long w = nsm; w; w &= w - 1) {
                    const int k = th_ctz64(w);
                    const int ck = TH_READLANE(cand, k);
                    if (nonskip && k < lane && pre0 == ck) trivial = true;
                }
            }
            const unsigned long long m = __ballot(nonskip && !trivial);
            const int f = th_ctz64(m);               // 64 when every live candidate of the window is trivial
            if (nonskip && trivial && lane < f) s_ft[cand] = (unsigned short)(w0 | 0x8000);      // (one 16-bit store per lane: the candidates of a window are distinct hits)
            TH_WAVE_FENCE();
            if (!m) { ci += TH_WAVE; continue; }
            const int c0 = TH_READLANE(cand, f);
            ci += f + 1;
            // ---- attempt: backtrack_dp into slot ch_n (wave-uniform scalar walk; cells written by lane 0) ----
            int sc = S[c0], cur = c0, len = 0;
            int wc = TH_READLANE(w0, f);             // the word of `cur`: it is not tracked yet
            for (;;) {
                if (lane == 0) { s_ft[cur] = (unsigned short)(wc | 0x8000); CL[top + len] = cur; }
                ++len;
                const int pre = (wc & 0x3fff) - 1;
                if (pre == -1) break;
                TH_WAVE_FENCE();
                const int wp = (int)s_ft[pre];       // the predecessor's flag AND its own predecessor: the only dependent read of the step
                if (wp >> 15) { sc -= S[pre]; break; }
                cur = pre; wc = wp;
                TH_WAVE_FENCE();     // every lane has read the word of `pre` before lane 0 rewrites it in the next trip
            }
            TH_WAVE_FENCE();
            const int nw = ch_n;
            const int cfirst = cur, clast = c0;      // the walk ran from the chain's last cell to its first
            int ok = 0;
            if (len > 1) {
                const int fs = E[cfirst] - Pd[cfirst], ls = E[clast] - Pd[clast], le = E[clast];
                int loses = 0, kill = -1;
                for (int i = ch_n - 1; i >= 0; --i) {            // is_overlap_chain, :68-83
                    if (ch_len.get(i) <= 0) continue;
                    if (ch_le.get(i) <= ls) break;
                    const int s1 = ch_fs.get(i), e1 = ch_ls.get(i);
                    const int mn = min(e1 - s1, ls - fs), ov = min(e1, ls) - max(s1, fs);
                    if (ov / (mn + 0.0) >= 0.5) {
                        if (ch_score.get(i) > sc) loses = 1; else kill = i;
                        break;
                    }
                }
                TH_WAVE_FENCE();
                if (lane == 0 && kill >= 0) ch_len.set(kill, 0);    // the older chain loses
                TH_WAVE_FENCE();
                ok = !loses;
                if (ok) {
                    // reverse the cells in place (they were written last-to-first)
                    for (int a = lane; a < len / 2; a += TH_WAVE) { const int b2 = len - 1 - a; const int t0 = CL[top + a], t1 = CL[top + b2]; CL[top + a] = t1; CL[top + b2] = t0; }
                    if (lane == 0) { ch_off.set(nw, top); ch_len.set(nw, len); ch_score.set(nw, sc); ch_fs.set(nw, fs); ch_ls.set(nw, ls); ch_le.set(nw, le); }
                    top += len; ++ch_n;
                }
            }
            TH_WAVE_FENCE();
            // sort_chain, src/tandem_chain.c:188-207 (selection pass with the reference's stale `i`), by lane 0
            if (lane == 0) {
                for (int a = 0; a + 1 < ch_n; ++a) {
                    const int i = ch_idx.get(a);
                    if (ch_len.get(i) <= 0) continue;
                    int e1 = ch_le.get(i);
                    for (int b = a + 1; b < ch_n; ++b) {
                        const int j = ch_idx.get(b);
                        if (ch_len.get(j) <= 0) continue;
                        const int e2 = ch_le.get(j);
                        if (e1 < e2) { ch_idx.set(a, j); ch_idx.set(b, i); e1 = e2; }
                    }
                }
            }
            TH_WAVE_FENCE();
        }
        // copy_chain in reverse rank order (ascending end), chains with >= 3 cells; seed = first cell (:251-255)
        int cnt = 0;
        for (int q = ch_n - 1; q >= 0; --q) if (ch_len.get(ch_idx.get(q)) >= 3) ++cnt;
        int base = 0;
        if (lane == 0 && cnt) base = atomicAdd(rec_count, cnt);
        base = TH_READLANE(base, 0);
        if (base + cnt > rec_cap) { if (lane == 0) atomicOr(err, 1); cnt = 0; }
        if (lane == 0) {
            read_ch_base[r] = base; read_ch_n[r] = cnt;
            if (cnt) {
                int o = base;
                for (int q = ch_n - 1; q >= 0; --q) {
                    const int c = ch_idx.get(q);
                    if (ch_len.get(c) < 3) continue;
                    const int cc0 = CL[ch_off.get(c)];
                    ThChainRec xr;
                    xr.read = r; xr.cell_off = (int)off + ch_off.get(c); xr.len = ch_len.get(c);
                    xr.est_start = ch_fs.get(c); xr.est_period = E[cc0] - ch_fs.get(c);
                    xr.first_end = E[cc0]; xr.last_start = ch_ls.get(c); xr.pad = 0;
                    rec[o++] = xr;
                }
            }
        }
        TH_WAVE_FENCE();
    }
}
