// th_ksw.h -- wave-cooperative affine-gap DP engine (gfx950) for the three ksw2 modes TideHunter uses.
//
// Replaces ksw_extz2_sse (ksw2/ksw2_extz2_sse.c:23-304) + ksw_backtrack (ksw2/ksw2.h:119-151) +
// ksw2_get_xid / ksw2_backtrack_left_end (src/ksw2_align.c:62-115) for the global mode, and the
// exact-max tracking of the extension mode (ksw2_extz2_sse.c:224-269, ksw2.h:160-176).
//
// Scoring is fixed by src/ksw2_align.c:11-17: match 1, mismatch -2, either base N(4) -> -1,
// gap of length l costs 2 + l, full matrix (w = -1), no z-drop.
//
// Layout: one alignment per wavefront.  Query columns are processed in strips of 64 (one column per
// lane); inside a strip the wave sweeps anti-diagonals: at step s lane l updates cell
// (row i = s - l, column j = j0 + l).  Values crossing to the right-hand neighbour move by one lane
// per step; the strip's last column is parked in a per-wave boundary buffer (coalesced 64-row
// chunks through an "outbox" register) and re-read by the next strip's lane 0 ("inbox").
//
// No traceback matrix is stored (ksw2 allocates (qlen+tlen-1)*n_col*16 bytes, 2 MB at 1000^2): what
// the callers need from the traceback are two path functionals, and both are propagated forward
// together with the scores, honouring ksw2's tie rules (diagonal over E over F; a gap is continued
// only if that is strictly better than re-opening it):
//   CH(i,j) = number of '=' columns on the path ksw_backtrack would follow from (i,j) in state H
//   T1(i,j) / T2(i,j) = the same when (i,j) is entered in the E / F state
//   RH / R1 / R2      = target bases the path consumes after (and including) the point where its
//                       last q_left_ext query bases begin (ksw2_backtrack_left_end)
#pragma once
#include "th_dev.h"

#define TH_KQ 2   // gap open
#define TH_KE 1   // gap extend

__device__ __forceinline__ int th_ksw_sc(int a, int b) { return (a == 4 || b == 4) ? -TH_KE : (a == b ? 1 : -2); }

// Global alignment.  q: query (columns), t: target (rows).  bnd: per-wave scratch of
// ((tlen+63)&~63) * (WITH_R ? 6 : 4) ints.  Returns iden_n; *tback = back-projection of q_left_ext.
// Every lane of the wave must call; results are wave-uniform.
template <bool WITH_R>
__device__ int th_ksw_global_wave(const uint8_t *__restrict__ q, int qlen, const uint8_t *__restrict__ t, int tlen,
                                  int q_left_ext, int *__restrict__ bnd, int *tback)
{
    const int lane = TH_LANE();
    const int NV = WITH_R ? 6 : 4;
    const int tpad = (tlen + 63) & ~63;
    const int jstar = qlen - q_left_ext;     // first column of the query suffix (WITH_R)
    int res_iden = 0, res_r = 0;
    for (int j0 = 0; j0 < qlen; j0 += TH_WAVE) {
        const int j = j0 + lane;
        const bool colok = j < qlen;
        const int qj = colok ? q[j] : 4;
        const bool first = j0 == 0, last = j0 + TH_WAVE >= qlen;
        // per-column state (own previous row)
        int Hup = 0, Ein = 0, T1up = 0, R1up = 0, Hdiag = 0, CHdiag = 0, R0diag = 0;
        // values produced by this lane at the previous step (consumed by lane+1)
        int oH = 0, oF = 0, oCH = 0, oT2 = 0, oR0 = 0, oR2 = 0, tcur = 4;
        // inbox / outbox chunks
        int ibH = 0, ibF = 0, ibCH = 0, ibT2 = 0, ibR0 = 0, ibR2 = 0;
        int nbH = 0, nbF = 0, nbCH = 0, nbT2 = 0, nbR0 = 0, nbR2 = 0;
        int obH = 0, obF = 0, obCH = 0, obT2 = 0, obR0 = 0, obR2 = 0;
        int tch = 4, tnx = 4;
        // prefetch chunk 0
        {
            const int row = lane;
            tnx = row < tlen ? t[row] : 4;
            if (!first && row < tlen) {
                nbH = bnd[0 * tpad + row]; nbF = bnd[1 * tpad + row]; nbCH = bnd[2 * tpad + row]; nbT2 = bnd[3 * tpad + row];
                if (WITH_R) { nbR0 = bnd[4 * tpad + row]; nbR2 = bnd[5 * tpad + row]; }
            }
        }
        const int nsteps = tlen + TH_WAVE - 1;
        for (int s = 0; s < nsteps; ++s) {
            if ((s & 63) == 0) {
                ibH = nbH; ibF = nbF; ibCH = nbCH; ibT2 = nbT2; ibR0 = nbR0; ibR2 = nbR2; tch = tnx;
                const int row = s + 64 + lane;
                tnx = row < tlen ? t[row] : 4;
                if (!first && row < tlen) {
                    nbH = bnd[0 * tpad + row]; nbF = bnd[1 * tpad + row]; nbCH = bnd[2 * tpad + row]; nbT2 = bnd[3 * tpad + row];
                    if (WITH_R) { nbR0 = bnd[4 * tpad + row]; nbR2 = bnd[5 * tpad + row]; }
                }
            }
            // left inputs: from lane-1 (previous step) or, for lane 0, from the inbox / the matrix's left border
            int lH = th_shr1(oH, 0), lF = th_shr1(oF, 0), lCH = th_shr1(oCH, 0), lT2 = th_shr1(oT2, 0);
            int lR0 = 0, lR2 = 0;
            if (WITH_R) { lR0 = th_shr1(oR0, 0); lR2 = th_shr1(oR2, 0); }
            int ti = th_shr1(tcur, 4);
            {
                const int sl = s & 63;
                const int bH = TH_READLANE(ibH, sl), bF = TH_READLANE(ibF, sl), bCH = TH_READLANE(ibCH, sl), bT2 = TH_READLANE(ibT2, sl);
                const int bt = TH_READLANE(tch, sl);
                int bR0 = 0, bR2 = 0;
                if (WITH_R) { bR0 = TH_READLANE(ibR0, sl); bR2 = TH_READLANE(ibR2, sl); }
                if (lane == 0) {
                    ti = bt;
                    if (first) { lH = -(TH_KQ + (s + 1) * TH_KE); lF = lH - TH_KQ - TH_KE; lCH = 0; lT2 = 0; lR0 = 0; lR2 = 0; }
                    else { lH = bH; lF = bF; lCH = bCH; lT2 = bT2; lR0 = bR0; lR2 = bR2; }
                }
            }
            tcur = ti;
            const int i = s - lane;
            const bool act = colok && i >= 0 && i < tlen;
            if (act) {
                if (i == 0) {   // top border of this column
                    Hup = -(TH_KQ + (j + 1) * TH_KE); Ein = Hup - TH_KQ - TH_KE; T1up = 0; R1up = tlen;
                    Hdiag = j == 0 ? 0 : -(TH_KQ + j * TH_KE); CHdiag = 0;
                    R0diag = tlen;   // a path that leaves through the top border still owes an insertion: all of t is consumed
                }
                const int diag = Hdiag + th_ksw_sc(ti, qj);
                int d = Ein > diag ? 1 : 0;
                int h = max(diag, Ein);
                if (lF > h) { d = 2; h = lF; }
                const bool xe = Ein > h - TH_KQ, xf = lF > h - TH_KQ;
                const int ch = d == 0 ? CHdiag + (ti == qj) : d == 1 ? T1up : lT2;
                const int t1 = xe ? T1up : ch, t2 = xf ? lT2 : ch;
                int r0 = 0, r1 = 0, r2 = 0;
                if (WITH_R) {
                    if (j == jstar) {
                        r0 = d == 0 ? tlen - i : d == 1 ? R1up : tlen - 1 - i;
                        r1 = xe ? R1up : r0;
                        r2 = xf ? tlen - 1 - i : r0;
                    } else {
                        r0 = d == 0 ? R0diag : d == 1 ? R1up : lR2;
                        r1 = xe ? R1up : r0;
                        r2 = xf ? lR2 : r0;
                    }
                }
                // roll the column state
                Hdiag = lH; CHdiag = lCH; R0diag = lR0;
                Hup = h; T1up = t1; R1up = r1;
                Ein = max(Ein, h - TH_KQ) - TH_KE;
                oH = h; oF = max(lF, h - TH_KQ) - TH_KE; oCH = ch; oT2 = t2; oR0 = r0; oR2 = r2;
                if (i == tlen - 1 && j == qlen - 1) { res_iden = ch; res_r = r0; }
            }
            // park the strip's last column for the next strip
            if (!last) {
                const int io = s - 63;                       // row lane 63 has just produced
                const int vH = TH_READLANE(oH, 63), vF = TH_READLANE(oF, 63), vCH = TH_READLANE(oCH, 63), vT2 = TH_READLANE(oT2, 63);
                int vR0 = 0, vR2 = 0;
                if (WITH_R) { vR0 = TH_READLANE(oR0, 63); vR2 = TH_READLANE(oR2, 63); }
                if (io >= 0 && io < tlen) {
                    if (lane == (io & 63)) { obH = vH; obF = vF; obCH = vCH; obT2 = vT2; obR0 = vR0; obR2 = vR2; }
                    if ((io & 63) == 63 || io == tlen - 1) {
                        const int row = (io & ~63) + lane;
                        if (row <= io) {
                            bnd[0 * tpad + row] = obH; bnd[1 * tpad + row] = obF; bnd[2 * tpad + row] = obCH; bnd[3 * tpad + row] = obT2;
                            if (WITH_R) { bnd[4 * tpad + row] = obR0; bnd[5 * tpad + row] = obR2; }
                        }
                    }
                }
            }
        }
        TH_WAVE_FENCE();
    }
    // the result sits in the lane that owns column qlen-1
    const int owner = (qlen - 1) & 63;
    res_iden = __shfl(res_iden, owner);
    if (WITH_R) res_r = __shfl(res_r, owner);
    if (tback) *tback = res_r;
    (void)NV;
    return res_iden;
}

// Extension, score only: returns (max_q, max_t) of the first strictly greater H over anti-diagonals
// r = 0,1,..., candidates inside one anti-diagonal ranked as ksw2's SSE code visits them
// (ksw2_extz2_sse.c:228-260): t = en0 first, then four interleaved lanes over [st0,en1), then the
// tail [en1,en0).  qs/ts = +1 or -1 walk q/t forwards or backwards (ksw2_left_ext reverses both,
// src/ksw2_align.c:161-173).  bnd: ((tlen+63)&~63) * 2 ints.
__device__ __forceinline__ int th_ext_rank(int r, int t, int qlen, int tlen)
{
    const int st0 = max(0, r - qlen + 1), en0 = min(r, tlen - 1);
    if (t == en0) return 0;
    const int en1 = st0 + ((en0 - st0) >> 2 << 2);
    if (t < en1) return 1 + ((t - st0) & 3) * 0x1000000 + ((t - st0) >> 2);
    return 1 + 4 * 0x1000000 + (t - en1);
}

__device__ void th_ksw_ext_wave(const uint8_t *__restrict__ q, int qs, int qlen, const uint8_t *__restrict__ t, int ts, int tlen,
                                int *__restrict__ bnd, int *max_q, int *max_t)
{
    const int lane = TH_LANE();
    const int tpad = (tlen + 63) & ~63;
    int bestH = 0, bestR = 0x7fffffff, bestRank = 0x7fffffff, bestT = -1;
    for (int j0 = 0; j0 < qlen; j0 += TH_WAVE) {
        const int j = j0 + lane;
        const bool colok = j < qlen;
        const int qj = colok ? q[(long long)j * qs] : 4;
        const bool first = j0 == 0, last = j0 + TH_WAVE >= qlen;
        int Hup = 0, Ein = 0, Hdiag = 0, oH = 0, oF = 0, tcur = 4;
        int ibH = 0, ibF = 0, nbH = 0, nbF = 0, obH = 0, obF = 0, tch = 4, tnx = 4;
        {
            const int row = lane;
            tnx = row < tlen ? t[(long long)row * ts] : 4;
            if (!first && row < tlen) { nbH = bnd[row]; nbF = bnd[tpad + row]; }
        }
        const int nsteps = tlen + TH_WAVE - 1;
        for (int s = 0; s < nsteps; ++s) {
            if ((s & 63) == 0) {
                ibH = nbH; ibF = nbF; tch = tnx;
                const int row = s + 64 + lane;
                tnx = row < tlen ? t[(long long)row * ts] : 4;
                if (!first && row < tlen) { nbH = bnd[row]; nbF = bnd[tpad + row]; }
            }
            int lH = th_shr1(oH, 0), lF = th_shr1(oF, 0), ti = th_shr1(tcur, 4);
            {
                const int sl = s & 63;
                const int bH = TH_READLANE(ibH, sl), bF = TH_READLANE(ibF, sl), bt = TH_READLANE(tch, sl);
                if (lane == 0) {
                    ti = bt;
                    if (first) { lH = -(TH_KQ + (s + 1) * TH_KE); lF = lH - TH_KQ - TH_KE; }
                    else { lH = bH; lF = bF; }
                }
            }
            tcur = ti;
            const int i = s - lane;
            const bool act = colok && i >= 0 && i < tlen;
            if (act) {
                if (i == 0) { Hup = -(TH_KQ + (j + 1) * TH_KE); Ein = Hup - TH_KQ - TH_KE; Hdiag = j == 0 ? 0 : -(TH_KQ + j * TH_KE); }
                int h = Hdiag + th_ksw_sc(ti, qj);
                h = max(h, Ein); h = max(h, lF);
                Hdiag = lH; Hup = h;
                Ein = max(Ein, h - TH_KQ) - TH_KE;
                oH = h; oF = max(lF, h - TH_KQ) - TH_KE;
                // within this lane r = i + j only grows, so a strict '>' keeps the earliest anti-diagonal
                if (h > bestH) { bestH = h; bestR = i + j; bestT = i; bestRank = th_ext_rank(i + j, i, qlen, tlen); }
            }
            if (!last) {
                const int io = s - 63;
                const int vH = TH_READLANE(oH, 63), vF = TH_READLANE(oF, 63);
                if (io >= 0 && io < tlen) {
                    if (lane == (io & 63)) { obH = vH; obF = vF; }
                    if ((io & 63) == 63 || io == tlen - 1) {
                        const int row = (io & ~63) + lane;
                        if (row <= io) { bnd[row] = obH; bnd[tpad + row] = obF; }
                    }
                }
            }
        }
        TH_WAVE_FENCE();
    }
    // wave arg-max by (H desc, anti-diagonal asc, SSE visiting rank asc)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int oH2 = __shfl_xor(bestH, d), oR = __shfl_xor(bestR, d), oK = __shfl_xor(bestRank, d), oT = __shfl_xor(bestT, d);
        const bool take = oH2 > bestH || (oH2 == bestH && (oR < bestR || (oR == bestR && oK < bestRank)));
        if (take) { bestH = oH2; bestR = oR; bestRank = oK; bestT = oT; }
    }
    if (bestH > 0) { *max_t = bestT; *max_q = bestR - bestT; } else { *max_t = -1; *max_q = -1; }
}
