// th_k_adapter.hip -- adapter search on the device (gfx950): edlib_align_HW as seqs_msa() calls it on the doubled consensus,
// src/gen_cons.c:226-291 through src/edlib_align.c:73-85 (edlib HW mode, EDLIB_TASK_LOC: edit distance <= k, the FIRST best end and
// the start that edlib reports for it, edlib/src/edlib.cpp:141-275,526-683).
//
// Myers' bit-vector recurrence, blocks of 64 query symbols (adapters of up to 256 symbols; longer ones stay on the host).  One
// search per LANE: job j runs the four searches the host logic may ask for (5' adapter, rc(3'), rc(5'), 3'), each forward over
// consensus + consensus for the distance and the first best end, then backward from that end with the reversed query in prefix
// mode for the start (the last best column, as edlib does).  Lanes of a wave hold different jobs; the work per lane is a few
// thousand steps of ~20 integer instructions per block, so no cross-lane cooperation is needed.  The equality tables (per
// consensus letter A,C,G,T,N: the query positions it matches, case-insensitively, src/edlib_align.c:21-27) come from the host.
#include "th_dev.h"

#define TH_AD_MAXB 4                       // 64-symbol blocks per query
struct ThAdPara {
    int len[4], k[4], nb[4];               // query length, edlib's k, blocks -- per search
    unsigned long long peq[4][2][5][TH_AD_MAXB];   // [search][0 forward / 1 reversed query][consensus letter][block]
};

template <bool HW>
__device__ __forceinline__ int th_myers_step(const unsigned long long *peq, int nb, int lastbit, unsigned long long *Pv, unsigned long long *Mv, int score)
{
    int hin = HW ? 0 : 1;
#pragma unroll
    for (int b = 0; b < TH_AD_MAXB; ++b) {
        if (b < nb) {
            unsigned long long Eq = peq[b];
            const unsigned long long pv = Pv[b], mv = Mv[b];
            const unsigned long long Xv = Eq | mv;
            if (hin < 0) Eq |= 1ull;
            const unsigned long long Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            unsigned long long Ph = mv | ~(Xh | pv), Mh = pv & Xh;
            int hout;
            if (b == nb - 1) { hout = (int)((Ph >> lastbit) & 1ull) - (int)((Mh >> lastbit) & 1ull); score += hout; }
            else hout = (int)(Ph >> 63) - (int)(Mh >> 63);
            Ph <<= 1; Mh <<= 1;
            if (hin < 0) Mh |= 1ull; else if (hin > 0) Ph |= 1ull;
            Pv[b] = Mh | ~(Xv | Ph); Mv[b] = Ph & Xv;
            hin = hout;
        }
    }
    return score;
}

// out[12 * job + 3 * search + {0,1,2}] = ed (-1: above k), start, end -- the values edlib_align_HW returns
__global__ void __launch_bounds__(64)
th_k_adapter(const ThTask *__restrict__ tasks, int n_tasks, const uint8_t *__restrict__ cons, const int *__restrict__ cons_len,
             const ThAdPara *__restrict__ AP, int min_cons, int *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int job = x >> 2, sr = x & 3;
    if (job >= n_tasks) return;
    int *o = out + 12 * job + 3 * sr;
    const int cl = cons_len[job], qlen = AP->len[sr];
    o[0] = -1; o[1] = -1; o[2] = -1;
    if (cl <= min_cons || qlen <= 0) return;                  // src/gen_cons.c:226: only consensuses longer than both adapters
    const uint8_t *c = cons + tasks[job].cons_off;
    const int tlen = cl << 1, nb = AP->nb[sr], lastbit = (qlen - 1) & 63, k = AP->k[sr];
    unsigned long long Pv[TH_AD_MAXB], Mv[TH_AD_MAXB];
#pragma unroll
    for (int b = 0; b < TH_AD_MAXB; ++b) { Pv[b] = ~0ull; Mv[b] = 0ull; }
    int score = qlen, best = 0x7fffffff, be = 0, p = 0;
    for (int t = 0; t < tlen; ++t) {
        const int code = min((int)c[p], 4);
        score = th_myers_step<true>(AP->peq[sr][0][code], nb, lastbit, Pv, Mv, score);
        if (score < best) { best = score; be = t; }
        if (++p == cl) p = 0;
    }
    if (k >= 0 && best > min(k, qlen)) return;                 // -1
    if (best >= qlen && (64 - qlen % 64) % 64 > 0) { o[0] = qlen; o[1] = 0; o[2] = -1; return; }     // edlib's "position -1"
#pragma unroll
    for (int b = 0; b < TH_AD_MAXB; ++b) { Pv[b] = ~0ull; Mv[b] = 0ull; }
    score = qlen;
    int last_p = -1;
    p = be % cl;
    for (int t = 0; t <= be; ++t) {
        const int code = min((int)c[p], 4);
        score = th_myers_step<false>(AP->peq[sr][1][code], nb, lastbit, Pv, Mv, score);
        if (score == best) last_p = t;
        if (--p < 0) p = cl - 1;
    }
    o[0] = best; o[1] = be - last_p; o[2] = be;
}
