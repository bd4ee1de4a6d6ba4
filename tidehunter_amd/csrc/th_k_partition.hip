// th_k_partition.hip -- stage 3: split the read into repeat units along one chain (gfx950).
//
// Replaces get_partition_pos_with_narrow_global_alignment(), src/partition.c:171-276.
// One chain per wavefront.  The seed unit is always the chain's first cell (est_ch_i == 0,
// src/tandem_chain.c:251-255), so the reference's left walk (:186-224, guarded by ch_i > 0) never
// runs and only the right walk (:236-273) is implemented.  Anchor search is a 64-wide scan (cell
// starts grow strictly along a chain); bridging alignments run on the wave DP engine with the
// forward back-projection functional instead of a CIGAR.
#include "th_dev.h"
#include "th_ksw.h"

#define TH_PART_BLOCK 64

// par: chain c writes at par + par_off[c], capacity par_cap[c]; bnd: per-wave scratch, bnd_stride ints each
__global__ void __launch_bounds__(TH_PART_BLOCK)
th_k_partition(const ThChainRec *__restrict__ rec, const int *__restrict__ n_rec_ptr, const uint32_t *__restrict__ roff, const uint8_t *__restrict__ bseq,
               const int *__restrict__ hit_end, const int *__restrict__ hit_per, const int *__restrict__ cells, ThDevPara P,
               int *__restrict__ par, int par_cap, int *__restrict__ par_top, int *__restrict__ par_off, int *__restrict__ par_n,
               int *__restrict__ bnd, long long bnd_stride, int *__restrict__ err, int cap_mul, int *__restrict__ queue)
{
    __shared__ int s_in[2][TH_WAVE][8];              // the packed strip engine's inbox (th_ksw_global_r_wave)
    const int lane = TH_LANE();
    const int wave = blockIdx.x;
    int *mybnd = bnd + (long long)wave * bnd_stride;
    const int k = P.k;
    if (TH_UNI(*err) & TH_E_REC_OVERFLOW) return;          // the record arena overflowed: the extraction is rerun with a larger one first
    const int n_rec = TH_UNI(*n_rec_ptr);                 // chain records of the batch (th_k_chain_extract*), read from the counter block
    for (;;) {
        const int c = th_queue_pop(queue);            // chains differ in length and in bridging alignments
        if (c >= n_rec) break;
        const ThChainRec R = rec[c];
        const uint32_t off = roff[R.read];
        const uint8_t *b = bseq + off;
        const int *E = hit_end + off, *Pd = hit_per + off, *CL = cells + R.cell_off;
        // the chain's par_pos region is carved from a shared arena (chain records arrive in no particular order,
        // so nothing derived from the record index may be used to keep regions apart)
        const int cap = cap_mul > 0 ? cap_mul * (2 * R.len + 16) : 4;      // (cap_mul 0: test knob, regions of four entries)
        int base = 0;
        { const int v = atomicAdd(par_top, lane == 0 ? cap : 0); base = TH_READLANE(v, 0); }
        int n = 0, bad = base + cap > par_cap;
        int *out = par + (bad ? 0 : base);
        if (lane == 0) par_off[c] = base;
        int ch_i = 0, s = R.est_start, e = R.est_start + R.est_period;
        if (lane == 0 && !bad) { out[0] = s; out[1] = e; }
        n = 2;
        while (ch_i < R.len - 1 && e <= R.last_start && !bad) {
            // first cell beyond ch_i whose start is >= e (exists: last_start >= e)
            int istar = -1;
            for (int base = ch_i + 1; base < R.len && istar < 0; base += TH_WAVE) {
                const int x = base + lane;
                int st = -1;
                if (x < R.len) { const int h = CL[x]; st = E[h] - Pd[h]; }
                const unsigned long long m = __ballot(x < R.len && st >= e);
                if (m) istar = base + th_ctz64(m);
            }
            if (istar < 0) break;   // cannot happen (see above)
            const int h2 = CL[istar];
            const int e2 = E[h2], s2 = e2 - Pd[h2];
            int s1 = s, e1 = e;
            if (istar - 1 > ch_i) { const int h1 = CL[istar - 1]; e1 = E[h1]; s1 = e1 - Pd[h1]; }
            if (n + 3 > cap) { bad = 1; break; }
            if (s2 == e) {                                   // abutting anchor, :244-247
                if (lane == 0) out[n] = e2;
                ++n; ch_i = istar; s = s2; e = e2;
            } else {                                         // bridge, :248-268
                const int ql = s2 - s1 + k, tl = e2 - e1 + k;
                if ((long long)((tl + 63) & ~63) * 6 > bnd_stride) { bad = 1; break; }
                int bt = 0;
                // (round 4: a certified band first -- two inter-anchor spans of a repeat differ by the read's error rate -- instead of the full matrix
                // in 32 bits on one wave, which at 5000-base periods was 50 ms during which the rest of the device had nothing to do)
                const int iden = th_ksw_global_r_wave(b + s1 - k + 1, ql, b + e1 - k + 1, tl, s2 - e, mybnd, s_in, &bt, !P.part_full);
                if ((double)iden >= (double)min(ql, tl) * P.one_minus_div) {
                    s = e; e = e2 - bt;
                    if (e == s) break;                       // no progress: ch_i = ch.len, :254-256
                    if (lane == 0) out[n] = e;
                    ++n; ch_i = istar - 1;
                } else {
                    if (lane == 0) { out[n] = -1; out[n + 1] = s2; out[n + 2] = e2; }
                    n += 3; ch_i = istar; s = s2; e = e2;
                }
            }
        }
        if (bad && lane == 0) atomicOr(err, 2);
        if (lane == 0) par_n[c] = bad ? 0 : n;
    }
}
