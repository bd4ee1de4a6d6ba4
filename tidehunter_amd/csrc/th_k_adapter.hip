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

// ---- `-s`: adapters searched in the WHOLE read (single_copy_full_len_seq / collect_ed_res, src/gen_cons.c:89-171) ----
// One lane per (read, query): edlib_align_HW over the read, then -- as collect_ed_res does -- once more over the part before the hit
// (if it starts at or beyond the query's length) and over the part from its end on (if that leaves room for the query).  The text is the
// read as it was uploaded (raw bytes), the equality tables cover all 256 byte values with the reference's rule (byte equality, a/A c/C
// g/G t/T n/N: src/edlib_align.c:21-27), built by the host.
struct ThAdRead { unsigned long long peq[4][2][256][TH_AD_MAXB]; };     // [query][0 forward / 1 reversed query][text byte][block]

__device__ __forceinline__ int th_edlib_hw_bytes(const unsigned long long (*pf)[TH_AD_MAXB], const unsigned long long (*pr)[TH_AD_MAXB], int nb, int qlen, int k,
                                                 const uint8_t *__restrict__ t, int tlen, int *start, int *end)
{
    if (qlen <= 0 || tlen <= 0) return -1;
    const int lastbit = (qlen - 1) & 63;
    unsigned long long Pv[TH_AD_MAXB], Mv[TH_AD_MAXB];
#pragma unroll
    for (int b = 0; b < TH_AD_MAXB; ++b) { Pv[b] = ~0ull; Mv[b] = 0ull; }
    int score = qlen, best = 0x7fffffff, be = 0;
    for (int c = 0; c < tlen; ++c) {
        score = th_myers_step<true>(pf[t[c]], nb, lastbit, Pv, Mv, score);
        if (score < best) { best = score; be = c; }
    }
    if (k >= 0 && best > min(k, qlen)) return -1;
    if (best >= qlen && (64 - qlen % 64) % 64 > 0) { *start = 0; *end = -1; return qlen; }     // edlib's "position -1"
#pragma unroll
    for (int b = 0; b < TH_AD_MAXB; ++b) { Pv[b] = ~0ull; Mv[b] = 0ull; }
    score = qlen;
    int last_p = -1;
    for (int c = 0; c <= be; ++c) {
        score = th_myers_step<false>(pr[t[be - c]], nb, lastbit, Pv, Mv, score);
        if (score == best) last_p = c;
    }
    *end = be; *start = be - last_p;
    return best;
}

// out[40 * read + 10 * query + 0] = number of hits (0..3), then (ed, start, end) of each -- what collect_ed_res leaves in its array
__global__ void __launch_bounds__(64)
th_k_adapter_reads(int n_reads, const uint32_t *__restrict__ roff, const uint8_t *__restrict__ seq, const ThAdPara *__restrict__ AP, const ThAdRead *__restrict__ AR, int *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = x >> 2, sr = x & 3;
    if (r >= n_reads) return;
    int *o = out + 40 * r + 10 * sr;
    const uint8_t *t = seq + roff[r]; const int len = (int)(roff[r + 1] - roff[r]);
    const int qlen = AP->len[sr], k = AP->k[sr], nb = AP->nb[sr];
    const unsigned long long (*pf)[TH_AD_MAXB] = AR->peq[sr][0], (*pr)[TH_AD_MAXB] = AR->peq[sr][1];
    int n = 0, s = 0, e = 0;
    int ed = th_edlib_hw_bytes(pf, pr, nb, qlen, k, t, len, &s, &e);
    if (ed != -1) {
        o[1] = ed; o[2] = s; o[3] = e; n = 1;
        const int s0 = s, e0 = e;
        if (s0 >= qlen) {
            ed = th_edlib_hw_bytes(pf, pr, nb, qlen, k, t, s0, &s, &e);
            if (ed != -1) { o[1 + 3 * n] = ed; o[2 + 3 * n] = s; o[3 + 3 * n] = e; ++n; }
        }
        if (e0 <= len - qlen) {
            ed = th_edlib_hw_bytes(pf, pr, nb, qlen, k, t + e0, len - e0, &s, &e);
            if (ed != -1) { o[1 + 3 * n] = ed; o[2 + 3 * n] = e0 + s; o[3 + 3 * n] = e0 + e; ++n; }
        }
    }
    o[0] = n;
}
