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
// (row i = s - l, column j = j0 + l).  Values crossing to the right-hand neighbour move one lane per
// step with a DPP wave shift whose lane-0 input is the strip's left border: that border sits in an
// "inbox" register holding 64 rows, rotated by one lane per step so the current row is always in
// lane 0.  Lane 63 parks its column (the next strip's left border) with one 16-byte store per step;
// the next strip prefetches it 64 rows at a time.  Lanes outside the matrix (before their first row,
// after their last, or beyond the last column) simply compute on: nothing they produce is consumed
// by a lane that is inside it, so the inner loop has no execution-mask juggling.
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

// The full-matrix engines are the last resort of every caller (a fraction of a per cent of the alignments): as functions of their own they
// stay out of the callers' register allocation and instruction stream (they touch global memory only, so nothing is lost by not inlining them)
#ifndef TH_KSW_FULL_ATTR
#define TH_KSW_FULL_ATTR TH_NOINLINE
#endif
#define TH_KQ 2   // gap open
#define TH_KE 1   // gap extend

__device__ __forceinline__ int th_ksw_sc(int a, int b) { return (a == 4 || b == 4) ? -TH_KE : (a == b ? 1 : -2); }

// Global alignment.  q: query (columns), t: target (rows).  bnd: per-wave scratch of
// ((tlen+63)&~63) * (WITH_R ? 6 : 4) ints (16-byte aligned).  Returns iden_n; *tback = back-projection
// of q_left_ext.  Every lane of the wave must call; results are wave-uniform.
template <bool WITH_R>
__device__ TH_KSW_FULL_ATTR int th_ksw_global_wave(const uint8_t *__restrict__ q, int qlen, const uint8_t *__restrict__ t, int tlen,
                                  int q_left_ext, int *__restrict__ bnd, int *tback)
{
    const int lane = TH_LANE();
    const int tpad = (tlen + 63) & ~63;
    int4 *bnd4 = (int4 *)bnd;                 // {H, F', CH, T2} of the parked column, one per row
    int2 *bndr = (int2 *)(bnd + 4 * tpad);    // {R0, R2} (WITH_R)
    const int jstar = qlen - q_left_ext;      // first column of the query suffix (WITH_R)
    const int nsteps = tlen + TH_WAVE - 1;
    int res_iden = 0, res_r = 0;
    for (int j0 = 0; j0 < qlen; j0 += TH_WAVE) {
        const int j = j0 + lane;
        const int qj = j < qlen ? q[j] : 4;
        const bool first = j0 == 0, last = j0 + TH_WAVE >= qlen;
        const bool owner = j == qlen - 1;
        int Hup = 0, Ein = 0, T1up = 0, R1up = 0, Hdiag = 0, CHdiag = 0, R0diag = 0;     // column state
        int oH = 0, oF = 0, oCH = 0, oT2 = 0, oR0 = 0, oR2 = 0, tcur = 4;                 // produced last step
        // next chunk of the left border (prefetched) and of the target
        int4 nb = { 0, 0, 0, 0 }; int2 nr = { 0, 0 }; int nt = 4;
        if (lane < tlen) { nt = t[lane]; if (!first) { nb = bnd4[lane]; if (WITH_R) nr = bndr[lane]; } }
        for (int s0 = 0; s0 < nsteps; s0 += TH_WAVE) {
            int ibH, ibF, ibCH, ibT2, ibR0 = 0, ibR2 = 0, ibT = nt;
            if (first) { ibH = -(TH_KQ + (s0 + lane + 1) * TH_KE); ibF = ibH - TH_KQ - TH_KE; ibCH = 0; ibT2 = 0; }
            else { ibH = nb.x; ibF = nb.y; ibCH = nb.z; ibT2 = nb.w; ibR0 = nr.x; ibR2 = nr.y; }
            {
                const int row = s0 + TH_WAVE + lane;
                nt = 4;
                if (row < tlen) { nt = t[row]; if (!first) { nb = bnd4[row]; if (WITH_R) nr = bndr[row]; } }
            }
            const int kend = min(TH_WAVE, nsteps - s0);
            for (int k = 0; k < kend; ++k) {
                const int i = s0 + k - lane;
                // left inputs: lane-1's outputs of the previous step; lane 0 takes the border row (inbox lane 0)
                const int lH = th_shr1_inject(oH, ibH), lF = th_shr1_inject(oF, ibF), lCH = th_shr1_inject(oCH, ibCH), lT2 = th_shr1_inject(oT2, ibT2);
                const int ti = th_shr1_inject(tcur, ibT);
                int lR0 = 0, lR2 = 0;
                if (WITH_R) { lR0 = th_shr1_inject(oR0, ibR0); lR2 = th_shr1_inject(oR2, ibR2); ibR0 = th_rol1(ibR0); ibR2 = th_rol1(ibR2); }
                ibH = th_rol1(ibH); ibF = th_rol1(ibF); ibCH = th_rol1(ibCH); ibT2 = th_rol1(ibT2); ibT = th_rol1(ibT);
                tcur = ti;
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
                Hdiag = lH; CHdiag = lCH; R0diag = lR0;
                Hup = h; T1up = t1; R1up = r1;
                Ein = max(Ein, h - TH_KQ) - TH_KE;
                oH = h; oF = max(lF, h - TH_KQ) - TH_KE; oCH = ch; oT2 = t2; oR0 = r0; oR2 = r2;
                if (owner && i == tlen - 1) { res_iden = ch; res_r = r0; }
                if (!last && lane == 63 && i >= 0 && i < tlen) {          // park the strip's last column
                    int4 o; o.x = oH; o.y = oF; o.z = oCH; o.w = oT2; bnd4[i] = o;
                    if (WITH_R) { int2 o2; o2.x = oR0; o2.y = oR2; bndr[i] = o2; }
                }
            }
        }
        TH_WAVE_FENCE();
    }
    const int ol = (qlen - 1) & 63;
    res_iden = TH_READLANE(res_iden, ol);
    if (WITH_R) res_r = TH_READLANE(res_r, ol);
    if (tback) *tback = res_r;
    return res_iden;
}

// Banded global alignment with an exactness certificate (iden_n only, no back-projection).
//
// Cells with diagonal d = i - j outside [lo, hi] = [min(0,D) - W, max(0,D) + W] (D = tlen - qlen) are
// treated as -infinity.  Any path that leaves the band must contain at least Gmin = 2W + 2 + |D| gap
// bases in at least two gap runs (one to get out, one to come back to the end corner), so it scores at
// most UB = (m + n - Gmin)/2 - 2*q - Gmin*e.  If the banded optimum S is strictly greater than UB,
// every optimal path of the full matrix -- and every candidate that could tie with it at a cell of
// the traceback -- lies inside the band, hence H, the tie-breaks and the '=' count along ksw_backtrack's
// path are those of the full computation.  Returns -1 when the certificate fails (the caller then
// runs the full matrix).  Strip geometry: columns j0..j0+63 need rows [j0+lo, j0+63+hi] only.
//
// slb > TH_NEG: a LOWER BOUND of the optimum (the banded score of an earlier pass).  Inside the band, every cell that no path of
// score >= slb can pass through is minus infinity as well: once the top border has left the band, paths enter a strip through the
// parked column only; a parked cell (i, j0 - 1) with H + min(a, b) - |a - b| e < slb (a = tlen - 1 - i, b = qlen - j0: what is left to
// the end corner; the path may cross INSIDE an open horizontal gap -- F == H at the parked cell -- and then closes the difference with
// |a - b| further extension bases and no new opening, so neither the opening nor an extra base may be charged: round 3 charged
// (|a - b| + 1) e and pruned the optimal path's own cell when slb was the optimum and the rest of the path all matches) is dead, and a
// live one reaches nothing that matters below row i + 64 + min(H + b - slb, (H + a - slb) / 2).
// Every optimal path, and every candidate that ties with one at a cell of the traceback, consists of live cells, so the result is
// the band's (th_ksw_ext_band_wave has the argument in full); the live region is a lens around the path, not a band.
__device__ int th_ksw_global_band_wave(const uint8_t *__restrict__ q, int qlen, const uint8_t *__restrict__ t, int tlen, int W, int *__restrict__ bnd, int (*s_in)[TH_WAVE][8], int *band_score, int slb = TH_NEG)
{
    const int lane = TH_LANE();
    const int D = tlen - qlen;
    const int lo = min(0, D) - W, hi = max(0, D) + W;
    int4 *bnd4 = (int4 *)bnd;
    int res_iden = 0, res_h = TH_NEG;
    int prs = 0, ppre = -1;                   // rows the previous strip swept (its parked column is valid there and nowhere else)
    for (int j0 = 0; j0 < qlen; j0 += TH_WAVE) {
        const int j = j0 + lane;
        const bool colok = j < qlen;
        const int qj = colok ? q[j] : 4;
        const bool first = j0 == 0, last = j0 + TH_WAVE >= qlen;
        const bool owner = j == qlen - 1;
        int rs = max(0, j0 + lo), re = min(tlen - 1, j0 + 63 + hi);           // rows of the band in this strip
        if (slb > TH_NEG && j0 + lo > 0) {                                     // (the top border is outside the band: entries through the parked column only)
            int vmin = 0x7fffffff, vmax = -1;
            const int b = qlen - j0;
            for (int r0 = prs; r0 <= ppre; r0 += TH_WAVE) {
                const int row = r0 + lane;
                if (row <= ppre) {
                    const int H = bnd4[row].x, a = tlen - 1 - row, dab = a > b ? a - b : b - a;
                    if (H + min(a, b) - dab * TH_KE >= slb) { vmin = min(vmin, row); vmax = max(vmax, row + 64 + min(H + b - slb, (H + a - slb) >> 1)); }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { vmin = min(vmin, __shfl_xor(vmin, d)); vmax = max(vmax, __shfl_xor(vmax, d)); }
            rs = max(rs, vmin); re = min(re, vmax);
        }
        rs = TH_UNI(rs); re = TH_UNI(re);
        if (rs > re) { *band_score = TH_NEG; return -1; }                      // (cannot happen with a valid lower bound: the optimal path is alive)
        const int nsteps = re - rs + TH_WAVE;
        int Hup = TH_NEG, Ein = TH_NEG, T1up = 0, Hdiag = TH_NEG, CHdiag = 0;
        int oH = TH_NEG, oF = TH_NEG, oCH = 0, oT2 = 0, tcur = 4;
        int4 nb = { TH_NEG, TH_NEG, 0, 0 }; int nt = 4;
        {
            const int row = rs + lane;
            if (row < tlen) { nt = t[row]; if (!first && row >= prs && row <= ppre) nb = bnd4[row]; }
        }
        for (int s0 = 0; s0 < nsteps; s0 += TH_WAVE) {
            // the 64 rows' left-border cells (previous strip's last column, or the matrix border) are parked in LDS, one
            // entry per row; every step all lanes read the entry of their step (one broadcast ds_read, issued a step
            // ahead) and wave_shr:1 injects it at lane 0 -- no rotating copy of the border in registers
            int (*inbox)[8] = s_in[(s0 >> 6) & 1];
            {
                int4 ib; const int ibT = nt;
                if (first) {
                    const int row = rs + s0 + lane;
                    const bool inb = row + 1 <= hi;                     // border column -1 has d = row + 1
                    ib.x = inb ? -(TH_KQ + (row + 1) * TH_KE) : TH_NEG; ib.y = inb ? ib.x - TH_KQ - TH_KE : TH_NEG; ib.z = 0; ib.w = 0;
                } else ib = nb;
                *(int4 *)&inbox[lane][0] = ib; inbox[lane][4] = ibT;
            }
            TH_WAVE_FENCE();
            {
                const int row = rs + s0 + TH_WAVE + lane;
                nt = 4; nb.x = TH_NEG; nb.y = TH_NEG; nb.z = 0; nb.w = 0;
                if (row < tlen) { nt = t[row]; if (!first && row >= prs && row <= ppre) nb = bnd4[row]; }
            }
            const int kend = min(TH_WAVE, nsteps - s0);
            int4 in = *(const int4 *)&inbox[0][0]; int inT = inbox[0][4];
            // one sweep step; `cin` / `cinT` hold this step's inbox entry, the next one is fetched into `nin` / `ninT`.
            // The loop below alternates two register sets so that no copy is needed between steps.
            const bool top_ok = rs + s0 < TH_WAVE;               // row 0 can only be met in a block that starts above row 64
            auto step = [&](int k, int4 &cin, int &cinT, int4 &nin, int &ninT) {
                const int i = rs + s0 + k - lane;
                const int lH = th_shr1_inject(oH, cin.x), lF = th_shr1_inject(oF, cin.y), lCH = th_shr1_inject(oCH, cin.z), lT2 = th_shr1_inject(oT2, cin.w);
                const int ti = th_shr1_inject(tcur, cinT);
                { const int kn = min(k + 1, TH_WAVE - 1); nin = *(const int4 *)&inbox[kn][0]; ninT = inbox[kn][4]; }
                tcur = ti;
                if (top_ok && i == 0) {   // top border (row -1) where it lies inside the band
                    Hup = (-1 - j >= lo) ? -(TH_KQ + (j + 1) * TH_KE) : TH_NEG; Ein = Hup > TH_NEG ? Hup - TH_KQ - TH_KE : TH_NEG; T1up = 0;
                    Hdiag = j == 0 ? 0 : ((-j >= lo) ? -(TH_KQ + j * TH_KE) : TH_NEG); CHdiag = 0;
                }
                const bool inb = colok && (unsigned)(i - rs) <= (unsigned)(re - rs) && (unsigned)(i - j - lo) <= (unsigned)(hi - lo);
                // ties prefer the diagonal over E over F (ksw2_extz2_sse.c:177-195); written as selects so that no lane diverges
                const int diag = Hdiag + th_ksw_sc(ti, qj);
                const bool e_gt = Ein > diag;
                const int h1 = max(diag, Ein);
                const bool f_gt = lF > h1;
                const int h = max(h1, lF);
                const bool xe = Ein > h - TH_KQ, xf = lF > h - TH_KQ;
                const int ch01 = e_gt ? T1up : CHdiag + (ti == qj ? 1 : 0);
                const int ch = f_gt ? lT2 : ch01;
                const int t1 = xe ? T1up : ch, t2 = xf ? lT2 : ch;
                Hdiag = lH; CHdiag = lCH;
                Hup = h; T1up = t1;
                Ein = inb ? max(Ein, h - TH_KQ) - TH_KE : TH_NEG;
                oH = inb ? h : TH_NEG; oF = inb ? max(lF, h - TH_KQ) - TH_KE : TH_NEG; oCH = ch; oT2 = t2;
                if (last && owner && i == tlen - 1) { res_iden = ch; res_h = h; }
                if (!last && lane == 63 && i >= rs && i <= re) { int4 o; o.x = oH; o.y = oF; o.z = oCH; o.w = oT2; bnd4[i] = o; }
            };
            int4 in2; int in2T;
            int k = 0;
            for (; k + 1 < kend; k += 2) { step(k, in, inT, in2, in2T); step(k + 1, in2, in2T, in, inT); }
            if (k < kend) step(k, in, inT, in2, in2T);
        }
        prs = rs; ppre = re;
        TH_WAVE_FENCE();
    }
    const int ol = (qlen - 1) & 63;
    res_iden = TH_READLANE(res_iden, ol); res_h = TH_READLANE(res_h, ol);
    const int gmin = 2 * W + 2 + (D < 0 ? -D : D);
    // S > (m + n - Gmin)/2 - 2q - Gmin*e   <=>   2S > (m + n - Gmin) - 4q - 2*Gmin*e
    *band_score = res_h;          // the banded optimum is a lower bound of the true optimum: it sizes a second, wider band
    if (2LL * res_h > (long long)(qlen + tlen - gmin) - 4 * TH_KQ - 2LL * gmin * TH_KE) return res_iden;
    return -1;
}

// ---- Banded global alignment, anti-diagonal formulation: the whole band of one anti-diagonal in one wave ----
//
// Same recurrence, tie rules, band semantics and certificate as th_ksw_global_band_wave above (so the two are
// interchangeable bit for bit), different geometry.  The strip sweep above spends a third of its lanes on the skew of
// every 64-column strip and moves five values one lane per step; here the wave holds one whole anti-diagonal
// r' = i' + j' of the band (i' = row + 1, j' = column + 1: row 0 / column 0 are the matrix borders) and walks r' = 0 ..
// tlen + qlen.  Cell position u on an anti-diagonal is its diagonal, d = i' - j' = lo + 2u + c with c = r' & 1 (lo is
// even), so that
//     the diagonal predecessor (i'-1, j'-1) is position u of anti-diagonal r'-2        -> stays in its register,
//     even r': the left neighbour is position u of r'-1, the upper one position u-1     -> one shift of (E', T1),
//     odd  r': the upper neighbour is position u of r'-1, the left one position u+1     -> one shift of (F', T2),
// and there is no skew, no parked strip border, no inbox.  Two positions share a lane as packed int16 halves
// (v_pk_add_i16 clamp / v_pk_max_i16 / v_pk_ashrrev_i16 / v_bfi, as in the POA rows): lane l, register k holds
// positions 2 (NP l + k) and the next one, so a shift by one position is one v_alignbit per register plus one DPP move
// per quantity.  NP registers cover diagonals lo .. lo + 256 NP - 1.  -32768 is minus infinity: every real value lies
// within +-(qlen + tlen + 8) (the caller guarantees qlen + tlen < 30000), saturating arithmetic keeps minus infinity
// below all of them, and positions outside the band or the matrix are reset to it after every step.
// Bases come from an LDS staging area (one byte per base): row codes at s_t[NPOS + i'], column codes REVERSED at
// s_q[qlen + NPOS - j'], both padded, so the two bases of a lane's pair are adjacent bytes for every r'.
// Steps come in three kinds: A (r' <= max(hi, -lo): border cells present, general validity mask), B (the band lies
// inside the matrix: constant masks), C (the rest: general mask).
#define TH_AD_ONE 0x00010001u
template <int NP> struct ThAdState {
    uint32_t E1[NP], TE[NP], F1[NP], TF[NP], Tp[NP], Qp[NP], NT[NP], NQ[NP], POS[NP];
};
__device__ __forceinline__ uint32_t th_ad_pair(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 16; }
__device__ __forceinline__ uint32_t th_ad_is4(uint32_t v) { return th_pk_negmask_op(th_pk_sub(v ^ 0x00040004u, TH_AD_ONE)); }

// one anti-diagonal.  C = r' & 1; PH = 0 (A) / 1 (B) / 2 (C); Hd / Cd: H and CH of anti-diagonal r'-2, replaced by those of r'
template <int NP, bool HASN, int C, int PH>
__device__ __forceinline__ void th_ad_step(ThAdState<NP> &S, uint32_t (&Hd)[NP], uint32_t (&Cd)[NP], const uint8_t *s_t, const uint8_t *s_q,
                                          int rp, int ib, int jb, int qlen, int tlen, int uhb)
{
    const int lane = TH_LANE();
    // bases: the sequence whose index advanced since r'-1 (rows on odd r', columns on even r')
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int u0 = 2 * (NP * lane + k);
        if (C) { S.Tp[k] = th_ad_pair(s_t + 128 * NP + ib + u0); if (HASN) S.NT[k] = th_ad_is4(S.Tp[k]); }
        else { S.Qp[k] = th_ad_pair(s_q + qlen + 128 * NP - jb + u0); if (HASN) S.NQ[k] = th_ad_is4(S.Qp[k]); }
    }
    uint32_t up[NP], upT[NP], lf[NP], lfT[NP];
    if (C == 0) {
        const uint32_t pe = (uint32_t)th_shr1((int)S.E1[NP - 1], (int)TH_PK_NEG), pt = (uint32_t)th_shr1_z((int)S.TE[NP - 1]);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            up[k] = th_alignbit16(S.E1[k], k ? S.E1[k - 1] : pe); upT[k] = th_alignbit16(S.TE[k], k ? S.TE[k - 1] : pt);
            lf[k] = S.F1[k]; lfT[k] = S.TF[k];
        }
    } else {
        const uint32_t nf = (uint32_t)th_shl1((int)S.F1[0], (int)TH_PK_NEG), nt = (uint32_t)th_shl1_z((int)S.TF[0]);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            lf[k] = th_alignbit16(k + 1 < NP ? S.F1[k + 1] : nf, S.F1[k]); lfT[k] = th_alignbit16(k + 1 < NP ? S.TF[k + 1] : nt, S.TF[k]);
            up[k] = S.E1[k]; upT[k] = S.TE[k];
        }
    }
    // validity of a position: inside the matrix (borders included) and inside the band
    uint32_t ULO = 0, UHI = 0, BT = 0, BL = 0, HBd = 0, EBd = 0;
    if (PH != 1) {
        const int ulo = max(0, max(-ib, jb - qlen)), uhi = min(min(tlen - ib, jb), uhb);
        ULO = TH_PK_DUP(ulo); UHI = TH_PK_DUP(uhi);
    }
    if (PH == 0) {
        const int ut = -ib >= 0 ? -ib : 0x7fff, ul = jb >= 0 ? jb : 0x7fff;     // positions of the border cells (0, r') and (r', 0)
        BT = TH_PK_DUP(ut); BL = TH_PK_DUP(ul);
        const int hb = rp == 0 ? 0 : -(TH_KQ + rp * TH_KE);
        HBd = TH_PK_DUP(hb); EBd = TH_PK_DUP(hb - TH_KQ - TH_KE);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const uint32_t EM = th_pk_negmask_op(th_pk_sub(S.Tp[k] ^ S.Qp[k], TH_AD_ONE));        // 0xffff where the bases are equal
        uint32_t sc = th_sel(EM, TH_AD_ONE, 0xfffefffeu);                                    // 1 / -2
        if (HASN) sc = th_sel(S.NT[k] | S.NQ[k], 0xffffffffu, sc);                           // either base N: -e
        const uint32_t diag = th_pk_add(Hd[k], sc);
        const uint32_t g_e = th_pk_negmask_op(th_pk_sub(diag, up[k]));                         // E > diagonal
        const uint32_t h1 = th_pk_max(diag, up[k]);
        const uint32_t g_f = th_pk_negmask_op(th_pk_sub(h1, lf[k]));                           // F > both
        uint32_t h = th_pk_max(h1, lf[k]);
        const uint32_t hq = th_pk_sub(h, TH_PK_DUP(TH_KQ));
        const uint32_t xe = th_pk_negmask_op(th_pk_sub(hq, up[k])), xf = th_pk_negmask_op(th_pk_sub(hq, lf[k]));
        uint32_t ch = th_sel(g_f, lfT[k], th_sel(g_e, upT[k], th_pk_sub(Cd[k], EM)));       // '=' count along ksw_backtrack's path
        uint32_t t1 = th_sel(xe, upT[k], ch), t2 = th_sel(xf, lfT[k], ch);
        uint32_t nE = th_pk_sub(th_pk_max(up[k], hq), TH_PK_DUP(TH_KE)), nF = th_pk_sub(th_pk_max(lf[k], hq), TH_PK_DUP(TH_KE));
        if (PH == 0) {          // border cells take their closed-form values
            const uint32_t BM = th_pk_negmask_op(th_pk_sub(S.POS[k] ^ BT, TH_AD_ONE)) | th_pk_negmask_op(th_pk_sub(S.POS[k] ^ BL, TH_AD_ONE));
            h = th_sel(BM, HBd, h); nE = th_sel(BM, EBd, nE); nF = th_sel(BM, EBd, nF);
            ch &= ~BM; t1 &= ~BM; t2 &= ~BM;
        }
        if (PH != 1) {      // phase B: the band fills the registers exactly (hi - lo = 256 NP - 1), every position is a cell of the band
            const uint32_t IM = th_pk_negmask_op(th_pk_sub(S.POS[k], ULO) | th_pk_sub(UHI, S.POS[k]));
            h = th_sel(IM, TH_PK_NEG, h); nE = th_sel(IM, TH_PK_NEG, nE); nF = th_sel(IM, TH_PK_NEG, nF);
        }
        Hd[k] = h; Cd[k] = ch; S.E1[k] = nE; S.TE[k] = t1; S.F1[k] = nF; S.TF[k] = t2;
    }
}

// two anti-diagonals (even r', then r' + 1)
template <int NP, bool HASN, int PH>
__device__ __forceinline__ void th_ad_pair_steps(ThAdState<NP> &S, uint32_t (&HA)[NP], uint32_t (&HB)[NP], uint32_t (&CA)[NP], uint32_t (&CB)[NP],
                                                const uint8_t *s_t, const uint8_t *s_q, int rp, int lo2, int qlen, int tlen, int uhb0, int uhb1)
{
    const int m = rp >> 1;
    th_ad_step<NP, HASN, 0, PH>(S, HA, CA, s_t, s_q, rp, m + lo2, m - lo2, qlen, tlen, uhb0);
    th_ad_step<NP, HASN, 1, PH>(S, HB, CB, s_t, s_q, rp + 1, m + 1 + lo2, m - lo2, qlen, tlen, uhb1);
}

// bases into the staging area; returns whether either sequence holds an N (code 4).  Needs tlen + qlen + 512 NP + 16 bytes.
__device__ __forceinline__ bool th_ad_stage(const uint8_t *__restrict__ q, int qlen, const uint8_t *__restrict__ t, int tlen, int npos, uint8_t *s, const uint8_t **st, const uint8_t **sq,
                                            int qs = 1, int ts = 1)      // qs / ts = +1 or -1: walk q / t forwards or backwards (left extension)
{
    const int lane = TH_LANE();
    const int ST = (tlen + 2 * npos + 8) & ~3, SQ = qlen + 2 * npos + 4;
    bool n4 = false;
#pragma unroll 4
    for (int x = lane; x < ST; x += TH_WAVE) { const int i = x - npos; uint8_t c = 4; if (i >= 1 && i <= tlen) { c = t[(long long)(i - 1) * ts]; n4 |= c == 4; } s[x] = c; }
#pragma unroll 4
    for (int y = lane; y < SQ; y += TH_WAVE) { const int j = qlen + npos - y; uint8_t c = 4; if (j >= 1 && j <= qlen) { c = q[(long long)(j - 1) * qs]; n4 |= c == 4; } s[ST + y] = c; }
    *st = s; *sq = s + ST;
    TH_WAVE_FENCE();
    return __ballot(n4) != 0;
}

// lo even, lo <= min(0, D), hi = lo + 256 NP - 1 >= max(0, D) (D = tlen - qlen); returns iden_n or -1 (not certified)
template <int NP, bool HASN>
__device__ __forceinline__ int th_ksw_band_ad_wave(const uint8_t *s_t, const uint8_t *s_q, int qlen, int tlen, int lo, int hi, int *band_score)
{
    const int lane = TH_LANE();
    const int lo2 = lo / 2, R = tlen + qlen, D = tlen - qlen;
    const int uhb0 = (hi - lo) >> 1, uhb1 = (hi - lo - 1) >> 1;
    ThAdState<NP> S;
    uint32_t HA[NP], HB[NP], CA[NP], CB[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int u0 = 2 * (NP * lane + k);
        HA[k] = HB[k] = TH_PK_NEG; CA[k] = CB[k] = 0; S.E1[k] = S.F1[k] = TH_PK_NEG; S.TE[k] = S.TF[k] = 0;
        S.POS[k] = TH_PK(u0, u0 + 1);
        S.Tp[k] = th_ad_pair(s_t + 128 * NP + lo2 + u0); S.Qp[k] = 0x00040004u;      // rows of r' = 0 (the even step loads its columns itself)
        S.NT[k] = HASN ? th_ad_is4(S.Tp[k]) : 0u; S.NQ[k] = 0u;
    }
    // phase limits (each phase runs whole pairs of anti-diagonals)
    const int endA = max(hi, -lo);                                        // last r' with a border cell
    const int limB = min(min(2 * tlen - hi, 2 * qlen + lo), R);           // last r' whose band lies inside the matrix
    int rp = 0;
    for (; rp <= endA && rp <= R; rp += 2) th_ad_pair_steps<NP, HASN, 0>(S, HA, HB, CA, CB, s_t, s_q, rp, lo2, qlen, tlen, uhb0, uhb1);
    for (; rp + 1 <= limB; rp += 2) th_ad_pair_steps<NP, HASN, 1>(S, HA, HB, CA, CB, s_t, s_q, rp, lo2, qlen, tlen, uhb0, uhb1);
    for (; rp <= R; rp += 2) th_ad_pair_steps<NP, HASN, 2>(S, HA, HB, CA, CB, s_t, s_q, rp, lo2, qlen, tlen, uhb0, uhb1);
    // cell (tlen, qlen) sits on anti-diagonal R at position tlen - ib(R)
    const int ibR = (R & 1) ? (R >> 1) + 1 + lo2 : (R >> 1) + lo2;
    const int us = tlen - ibR, ps = us >> 1, ks = ps % NP, ls = ps / NP;
    uint32_t hv = 0, cv = 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) if (k == ks) { hv = (R & 1) ? HB[k] : HA[k]; cv = (R & 1) ? CB[k] : CA[k]; }
    hv = (uint32_t)TH_READLANE((int)hv, ls); cv = (uint32_t)TH_READLANE((int)cv, ls);
    const int res_h = (us & 1) ? TH_PK_HI(hv) : TH_PK_LO(hv), res_iden = (us & 1) ? TH_PK_HI(cv) : TH_PK_LO(cv);
    *band_score = res_h;
    if (lo <= -qlen && hi >= tlen) return res_iden;                      // the band is the whole matrix
    const int gmin = min(2 * hi + 2 - D, -2 * lo + 2 + D);
    if (2LL * res_h > (long long)(qlen + tlen - gmin) - 4 * TH_KQ - 2LL * gmin * TH_KE) return res_iden;
    return -1;
}

// ---- Extension (score only) on the anti-diagonal engine ----
// Same geometry as th_ksw_band_ad_wave with the band [lo, hi] = [-128 NP, 128 NP - 1] around the main diagonal, the recurrence of
// th_ksw_ext_wave below (no path functionals), and per position a running best (H, anti-diagonal): within one position the
// anti-diagonal only grows, so a strict '>' keeps the earliest one; the final arg-max ranks candidates by (H desc, anti-diagonal
// asc, ksw2's SSE visiting order asc) exactly as th_ksw_ext_wave does.  Exactness certificate: a path that touches a diagonal d
// outside the band (d = i' - j', t along i) holds at least |d| gap bases (q + |d| e) AND has given up the diagonal steps the
// shorter remaining sequence no longer offers -- at most min(tlen - d, qlen) of them in all for d > 0, min(qlen + d, tlen) for
// d < 0 -- so every cell whose true H differs from the banded one, and every cell outside the band, has H <= th_adx_bound(); if the
// banded maximum S is above that bound (or the band is the whole matrix, or neither S nor the bound is positive) the result is
// ksw2's.  (Round 2 counted the gap bases only: min(qlen, tlen) - 3 - W, twice the band for sequences of similar length.)
__device__ __forceinline__ int th_adx_bound(int np, int qlen, int tlen)
{
    const int dt = 128 * np, dq = 128 * np + 1;            // the first diagonals outside [-128 NP, 128 NP - 1] on the t and on the q side
    const int bt = min(tlen - dt, qlen) - TH_KQ - dt * TH_KE, bq = min(qlen - dq, tlen) - TH_KQ - dq * TH_KE;
    return max(bt, bq);
}
template <int NP> struct ThAdxState { uint32_t E1[NP], F1[NP], Tp[NP], Qp[NP], NT[NP], NQ[NP], POS[NP], BH[NP], BR[NP]; };

template <int NP, bool HASN, int C, int PH>
__device__ __forceinline__ void th_adx_step(ThAdxState<NP> &S, uint32_t (&Hd)[NP], const uint8_t *s_t, const uint8_t *s_q,
                                           int rp, int ib, int jb, int qlen, int tlen, int uhb)
{
    const int lane = TH_LANE();
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int u0 = 2 * (NP * lane + k);
        if (C) { S.Tp[k] = th_ad_pair(s_t + 128 * NP + ib + u0); if (HASN) S.NT[k] = th_ad_is4(S.Tp[k]); }
        else { S.Qp[k] = th_ad_pair(s_q + qlen + 128 * NP - jb + u0); if (HASN) S.NQ[k] = th_ad_is4(S.Qp[k]); }
    }
    uint32_t up[NP], lf[NP];
    if (C == 0) {
        const uint32_t pe = (uint32_t)th_shr1((int)S.E1[NP - 1], (int)TH_PK_NEG);
#pragma unroll
        for (int k = 0; k < NP; ++k) { up[k] = th_alignbit16(S.E1[k], k ? S.E1[k - 1] : pe); lf[k] = S.F1[k]; }
    } else {
        const uint32_t nf = (uint32_t)th_shl1((int)S.F1[0], (int)TH_PK_NEG);
#pragma unroll
        for (int k = 0; k < NP; ++k) { lf[k] = th_alignbit16(k + 1 < NP ? S.F1[k + 1] : nf, S.F1[k]); up[k] = S.E1[k]; }
    }
    uint32_t ULO = 0, UHI = 0, BT = 0, BL = 0, HBd = 0, EBd = 0;
    if (PH != 1) {
        const int ulo = max(0, max(-ib, jb - qlen)), uhi = min(min(tlen - ib, jb), uhb);
        ULO = TH_PK_DUP(ulo); UHI = TH_PK_DUP(uhi);
    }
    if (PH == 0) {
        const int ut = -ib >= 0 ? -ib : 0x7fff, ul = jb >= 0 ? jb : 0x7fff;
        BT = TH_PK_DUP(ut); BL = TH_PK_DUP(ul);
        const int hb = rp == 0 ? 0 : -(TH_KQ + rp * TH_KE);
        HBd = TH_PK_DUP(hb); EBd = TH_PK_DUP(hb - TH_KQ - TH_KE);
    }
    const uint32_t RP = TH_PK_DUP(rp);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const uint32_t EM = th_pk_negmask_op(th_pk_sub(S.Tp[k] ^ S.Qp[k], TH_AD_ONE));
        uint32_t sc = th_sel(EM, TH_AD_ONE, 0xfffefffeu);
        if (HASN) sc = th_sel(S.NT[k] | S.NQ[k], 0xffffffffu, sc);
        uint32_t h = th_pk_max(th_pk_max(th_pk_add(Hd[k], sc), up[k]), lf[k]);
        const uint32_t hq = th_pk_sub(h, TH_PK_DUP(TH_KQ));
        uint32_t nE = th_pk_sub(th_pk_max(up[k], hq), TH_PK_DUP(TH_KE)), nF = th_pk_sub(th_pk_max(lf[k], hq), TH_PK_DUP(TH_KE));
        if (PH == 0) {
            const uint32_t BM = th_pk_negmask_op(th_pk_sub(S.POS[k] ^ BT, TH_AD_ONE)) | th_pk_negmask_op(th_pk_sub(S.POS[k] ^ BL, TH_AD_ONE));
            h = th_sel(BM, HBd, h); nE = th_sel(BM, EBd, nE); nF = th_sel(BM, EBd, nF);
        }
        if (PH != 1) {
            const uint32_t IM = th_pk_negmask_op(th_pk_sub(S.POS[k], ULO) | th_pk_sub(UHI, S.POS[k]));
            h = th_sel(IM, TH_PK_NEG, h); nE = th_sel(IM, TH_PK_NEG, nE); nF = th_sel(IM, TH_PK_NEG, nF);
        }
        // border cells are <= 0 and cells outside the matrix or the band are minus infinity: neither can beat a best that starts at 0
        const uint32_t upd = th_pk_negmask_op(th_pk_sub(S.BH[k], h));
        S.BH[k] = th_pk_max(S.BH[k], h); S.BR[k] = th_sel(upd, RP, S.BR[k]);
        Hd[k] = h; S.E1[k] = nE; S.F1[k] = nF;
    }
}

template <int NP, bool HASN, int PH>
__device__ __forceinline__ void th_adx_pair_steps(ThAdxState<NP> &S, uint32_t (&HA)[NP], uint32_t (&HB)[NP], const uint8_t *s_t, const uint8_t *s_q,
                                                 int rp, int lo2, int qlen, int tlen, int uhb0, int uhb1)
{
    const int m = rp >> 1;
    th_adx_step<NP, HASN, 0, PH>(S, HA, s_t, s_q, rp, m + lo2, m - lo2, qlen, tlen, uhb0);
    th_adx_step<NP, HASN, 1, PH>(S, HB, s_t, s_q, rp + 1, m + 1 + lo2, m - lo2, qlen, tlen, uhb1);
}

__device__ __forceinline__ int th_ext_rank(int r, int t, int qlen, int tlen);

// returns true when the result is certified; (max_q, max_t) as th_ksw_ext_wave; *best_h: the banded maximum (a lower bound of the
// true one: the smallest band whose th_adx_bound() lies below it is certified for sure)
template <int NP, bool HASN>
__device__ __forceinline__ bool th_ksw_ext_ad_wave(const uint8_t *s_t, const uint8_t *s_q, int qlen, int tlen, int *max_q, int *max_t, int *best_h)
{
    const int lane = TH_LANE();
    const int lo = -128 * NP, hi = 128 * NP - 1, lo2 = lo / 2, R = tlen + qlen;
    const int uhb0 = (hi - lo) >> 1, uhb1 = (hi - lo - 1) >> 1;
    ThAdxState<NP> S;
    uint32_t HA[NP], HB[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int u0 = 2 * (NP * lane + k);
        HA[k] = HB[k] = TH_PK_NEG; S.E1[k] = S.F1[k] = TH_PK_NEG; S.BH[k] = 0; S.BR[k] = 0;
        S.POS[k] = TH_PK(u0, u0 + 1);
        S.Tp[k] = th_ad_pair(s_t + 128 * NP + lo2 + u0); S.Qp[k] = 0x00040004u;
        S.NT[k] = HASN ? th_ad_is4(S.Tp[k]) : 0u; S.NQ[k] = 0u;
    }
    const int endA = max(hi, -lo);
    const int limB = min(min(2 * tlen - hi, 2 * qlen + lo), R);
    // Round 6: the band follows the MAIN diagonal, so it leaves the matrix where the shorter sequence ends -- position u of anti-diagonal pair m = rp / 2 is the cell
    // (i', j') = (m + lo2 + u, m - lo2 - u), 0 <= u < 128 NP, and none of them has j' <= qlen once m > qlen + 64 NP, none i' <= tlen once m > tlen + 64 NP.  Every later
    // anti-diagonal holds minus infinity only (the third loop's mask) and cannot move a best that starts at 0: the sweep ends there instead of at qlen + tlen.  The
    // headline shape's right flank is what the partition left of the read -- ~3000 bases against a 1000-base consensus: 2130 anti-diagonals instead of 3970
    // (profiles/r06_align_phases.json: the extensions were 18 % of the launch)
    const int Rend = min(R, 2 * min(qlen, tlen) + 128 * NP + 4);
    int rp = 0;
    for (; rp <= endA && rp <= Rend; rp += 2) th_adx_pair_steps<NP, HASN, 0>(S, HA, HB, s_t, s_q, rp, lo2, qlen, tlen, uhb0, uhb1);
    for (; rp + 1 <= limB; rp += 2) th_adx_pair_steps<NP, HASN, 1>(S, HA, HB, s_t, s_q, rp, lo2, qlen, tlen, uhb0, uhb1);
    for (; rp <= Rend; rp += 2) th_adx_pair_steps<NP, HASN, 2>(S, HA, HB, s_t, s_q, rp, lo2, qlen, tlen, uhb0, uhb1);
    // this lane's best candidate, then the wave's: (H desc, anti-diagonal asc, SSE visiting rank asc)
    int bestH = 0, bestR = 0x7fffffff, bestT = -1, bestRank = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int H = hh ? TH_PK_HI(S.BH[k]) : TH_PK_LO(S.BH[k]);
            const int rq = hh ? TH_PK_HI(S.BR[k]) : TH_PK_LO(S.BR[k]);       // r' = i' + j' of the best cell of this position
            if (H > 0) {
                const int u = 2 * (NP * lane + k) + hh;
                const int ibr = (rq & 1) ? (rq >> 1) + 1 + lo2 : (rq >> 1) + lo2;
                const int t = ibr + u - 1, r = rq - 2;                         // 0-based row and ksw2's anti-diagonal
                const int rk = th_ext_rank(r, t, qlen, tlen);
                const bool take = H > bestH || (H == bestH && (r < bestR || (r == bestR && rk < bestRank)));
                if (take) { bestH = H; bestR = r; bestT = t; bestRank = rk; }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int oH2 = __shfl_xor(bestH, d), oR = __shfl_xor(bestR, d), oK = __shfl_xor(bestRank, d), oT = __shfl_xor(bestT, d);
        const bool take = oH2 > bestH || (oH2 == bestH && (oR < bestR || (oR == bestR && oK < bestRank)));
        if (take) { bestH = oH2; bestR = oR; bestRank = oK; bestT = oT; }
    }
    if (bestH > 0) { *max_t = bestT; *max_q = bestR - bestT; } else { *max_t = -1; *max_q = -1; }
    const int bound = th_adx_bound(NP, qlen, tlen);
    *best_h = bestH;
    return (lo <= -qlen && hi >= tlen) || bestH > bound || (bestH <= 0 && bound <= 0);
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

__device__ TH_KSW_FULL_ATTR void th_ksw_ext_wave(const uint8_t *__restrict__ q, int qs, int qlen, const uint8_t *__restrict__ t, int ts, int tlen,
                                int *__restrict__ bnd, int *max_q, int *max_t)
{
    const int lane = TH_LANE();
    int2 *bnd2 = (int2 *)bnd;                 // {H, F'} of the parked column
    const int nsteps = tlen + TH_WAVE - 1;
    int bestH = 0, bestR = 0x7fffffff, bestT = -1;
    for (int j0 = 0; j0 < qlen; j0 += TH_WAVE) {
        const int j = j0 + lane;
        const bool colok = j < qlen;
        const int qj = colok ? q[(long long)j * qs] : 4;
        const bool first = j0 == 0, last = j0 + TH_WAVE >= qlen;
        int Ein = 0, Hdiag = 0, oH = 0, oF = 0, tcur = 4;
        int lbH = bestH - 1, lbT = -1;            // this strip's best at or above the running one (see the merge below)
        int2 nb = { 0, 0 }; int nt = 4;
        if (lane < tlen) { nt = t[(long long)lane * ts]; if (!first) nb = bnd2[lane]; }
        for (int s0 = 0; s0 < nsteps; s0 += TH_WAVE) {
            int ibH, ibF, ibT = nt;
            if (first) { ibH = -(TH_KQ + (s0 + lane + 1) * TH_KE); ibF = ibH - TH_KQ - TH_KE; }
            else { ibH = nb.x; ibF = nb.y; }
            {
                const int row = s0 + TH_WAVE + lane;
                nt = 4;
                if (row < tlen) { nt = t[(long long)row * ts]; if (!first) nb = bnd2[row]; }
            }
            const int kend = min(TH_WAVE, nsteps - s0);
            for (int k = 0; k < kend; ++k) {
                const int i = s0 + k - lane;
                const int lH = th_shr1_inject(oH, ibH), lF = th_shr1_inject(oF, ibF), ti = th_shr1_inject(tcur, ibT);
                ibH = th_rol1(ibH); ibF = th_rol1(ibF); ibT = th_rol1(ibT);
                tcur = ti;
                if (i == 0) { const int hu = -(TH_KQ + (j + 1) * TH_KE); Ein = hu - TH_KQ - TH_KE; Hdiag = j == 0 ? 0 : -(TH_KQ + j * TH_KE); }
                int h = Hdiag + th_ksw_sc(ti, qj);
                h = max(h, Ein); h = max(h, lF);
                Hdiag = lH;
                Ein = max(Ein, h - TH_KQ) - TH_KE;
                oH = h; oF = max(lF, h - TH_KQ) - TH_KE;
                // within one strip this lane's r = i + j only grows, so a strict '>' keeps the earliest anti-diagonal of the strip
                if (colok && i >= 0 && i < tlen && h > lbH) { lbH = h; lbT = i; }
                if (!last && lane == 63 && i >= 0 && i < tlen) { int2 o; o.x = oH; o.y = oF; bnd2[i] = o; }
            }
        }
        // across strips the lane's columns grow by 64 while the rows start over: an equal score on an EARLIER anti-diagonal may turn up in
        // a later strip, and ksw2 reports the earliest
        // (the same score on the SAME anti-diagonal, 64 k columns apart: ksw2's visiting order decides, as in the final reduction)
        if (lbT >= 0 && (lbH > bestH || lbT + j < bestR || (lbT + j == bestR && th_ext_rank(bestR, lbT, qlen, tlen) < th_ext_rank(bestR, bestT, qlen, tlen)))) { bestH = lbH; bestR = lbT + j; bestT = lbT; }
        TH_WAVE_FENCE();
    }
    // wave arg-max by (H desc, anti-diagonal asc, SSE visiting rank asc)
    int bestRank = bestT >= 0 ? th_ext_rank(bestR, bestT, qlen, tlen) : 0x7fffffff;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int oH2 = __shfl_xor(bestH, d), oR = __shfl_xor(bestR, d), oK = __shfl_xor(bestRank, d), oT = __shfl_xor(bestT, d);
        const bool take = oH2 > bestH || (oH2 == bestH && (oR < bestR || (oR == bestR && oK < bestRank)));
        if (take) { bestH = oH2; bestR = oR; bestRank = oK; bestT = oT; }
    }
    if (bestH > 0) { *max_t = bestT; *max_q = bestR - bestT; } else { *max_t = -1; *max_q = -1; }
}

// The bound of th_adx_bound() for the symmetric band |i - j| <= W of the strip engine below.
__device__ __forceinline__ int th_ext_bound_w(int W, int qlen, int tlen)
{
    const int d = W + 1;
    return max(min(tlen - d, qlen), min(qlen - d, tlen)) - TH_KQ - d * TH_KE;
}

// th_ksw_ext_wave with everything pruned that cannot matter, given a LOWER BOUND slb of the maximum (a banded maximum of the
// anti-diagonal engine: any banded result is one).  A cell can only matter -- be the maximum, tie with it, or lie on a path to such a
// cell -- if its H plus the diagonal steps still possible after it reaches slb; cells that cannot are minus infinity.  Two prunings
// follow, both exact for any slb <= the true maximum (no certificate, no retry):
//   * the diagonals |i - j| <= W, W chosen by the caller with th_ext_bound_w(W) < slb (the matrix borders keep their exact values,
//     which only adds real paths): strip j0 sweeps rows max(0, j0 - W) .. min(tlen - 1, j0 + 63 + W) at most;
//   * once the top border has left that band, paths enter a strip through the parked column only: a parked cell (i, j0 - 1) with
//     H + min(A, B) < slb (A = tlen - 1 - i, B = qlen - j0 diagonal steps left) is dead, and a live one reaches no cell that matters
//     below row i + 64 + min(H + B - slb, (H + A - slb) / 2) (a - c surplus rows cost a gap base each and, where t binds, a diagonal
//     step each).  The live rows of a repeat's flank shrink towards the end of the matrix, to about a quarter of its cells at 15 %
//     divergence where the band alone keeps half.
__device__ void th_ksw_ext_band_wave(const uint8_t *__restrict__ q, int qs, int qlen, const uint8_t *__restrict__ t, int ts, int tlen, int W, int slb,
                                     int *__restrict__ bnd, int (*s_in)[TH_WAVE][8], int *max_q, int *max_t)
{
    const int lane = TH_LANE();
    int2 *bnd2 = (int2 *)bnd;                 // {H, F'} of the parked column
    int bestH = 0, bestR = 0x7fffffff, bestT = -1;
    int prs = 0, ppre = -1;                   // rows the previous strip swept (its parked column is valid there and nowhere else)
    for (int j0 = 0; j0 < qlen; j0 += TH_WAVE) {
        const int j = j0 + lane;
        const bool colok = j < qlen;
        const int qj = colok ? q[(long long)j * qs] : 4;
        const bool first = j0 == 0, last = j0 + TH_WAVE >= qlen;
        int rs = max(0, j0 - W), re = min(tlen - 1, j0 + 63 + W);             // rows of the band in this strip
        if (j0 > W) {                                                          // the top border is outside the band: entries through the parked column only
            int vmin = 0x7fffffff, vmax = -1;
            const int B = qlen - j0;
            for (int r0 = prs; r0 <= ppre; r0 += TH_WAVE) {
                const int row = r0 + lane;
                if (row <= ppre) {
                    const int H = bnd2[row].x, A = tlen - 1 - row;
                    if (H + min(A, B) >= slb) { vmin = min(vmin, row); vmax = max(vmax, row + 64 + min(H + B - slb, (H + A - slb) >> 1)); }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { vmin = min(vmin, __shfl_xor(vmin, d)); vmax = max(vmax, __shfl_xor(vmax, d)); }
            rs = max(rs, vmin); re = min(re, vmax);
        }
        rs = TH_UNI(rs); re = TH_UNI(re);
        if (rs > re) break;                                                    // nothing alive crosses into this strip: nor into any later one
        const int nsteps = re - rs + TH_WAVE;
        const int jmW = colok ? j - W : -0x40000000;                           // in-band test: 0 <= i - jmW <= 2 W (never true beyond the last column)
        const unsigned rr = (unsigned)(re - rs), w2 = (unsigned)(2 * W);
        int Ein = TH_NEG, Hdiag = TH_NEG, oH = TH_NEG, oF = TH_NEG, tcur = 4;
        int lbH = bestH - 1, lbT = -1;                                         // this strip's best at or above the running one
        int2 nb = { TH_NEG, TH_NEG }; int nt = 4;
        { const int row = rs + lane; if (row < tlen) { nt = t[(long long)row * ts]; if (!first && row >= prs && row <= ppre) nb = bnd2[row]; } }
        for (int s0 = 0; s0 < nsteps; s0 += TH_WAVE) {
            // the 64 rows' left-border cells and their t bases sit in LDS, one entry per row; every step all lanes read the entry of their
            // step (one broadcast ds_read, issued a step ahead) and wave_shr:1 injects it at lane 0 (as th_ksw_global_band_wave)
            int (*inbox)[8] = s_in[(s0 >> 6) & 1];
            {
                int4 ib;
                if (first) { ib.x = -(TH_KQ + (rs + s0 + lane + 1) * TH_KE); ib.y = ib.x - TH_KQ - TH_KE; }
                else { ib.x = nb.x; ib.y = nb.y; }
                ib.z = nt; ib.w = 0;
                *(int4 *)&inbox[lane][0] = ib;
            }
            TH_WAVE_FENCE();
            {
                const int row = rs + s0 + TH_WAVE + lane;
                nt = 4; nb.x = TH_NEG; nb.y = TH_NEG;
                if (row < tlen) { nt = t[(long long)row * ts]; if (!first && row >= prs && row <= ppre) nb = bnd2[row]; }
            }
            const int kend = min(TH_WAVE, nsteps - s0);
            const bool top_ok = rs + s0 < TH_WAVE;               // row 0 can only be met in a block that starts above row 64
            int4 in = *(const int4 *)&inbox[0][0];
            auto step = [&](int k, int4 &cin, int4 &nin) {
                const int i = rs + s0 + k - lane;
                const int lH = th_shr1_inject(oH, cin.x), lF = th_shr1_inject(oF, cin.y), ti = th_shr1_inject(tcur, cin.z);
                { const int kn = min(k + 1, TH_WAVE - 1); nin = *(const int4 *)&inbox[kn][0]; }
                tcur = ti;
                if (top_ok && i == 0) { const int hu = -(TH_KQ + (j + 1) * TH_KE); Ein = hu - TH_KQ - TH_KE; Hdiag = j == 0 ? 0 : -(TH_KQ + j * TH_KE); }
                const bool inr = (unsigned)(i - rs) <= rr;
                const bool inb = inr && (unsigned)(i - jmW) <= w2;
                int h = Hdiag + th_ksw_sc(ti, qj);
                h = max(h, Ein); h = max(h, lF);
                Hdiag = lH;
                Ein = inb ? max(Ein, h - TH_KQ) - TH_KE : TH_NEG;
                oH = inb ? h : TH_NEG; oF = inb ? max(lF, h - TH_KQ) - TH_KE : TH_NEG;
                if (inb && h > lbH) { lbH = h; lbT = i; }       // within a strip this lane's anti-diagonal only grows: '>' keeps the earliest
                if (!last && lane == 63 && inr) { int2 o; o.x = oH; o.y = oF; bnd2[i] = o; }
            };
            int4 in2;
            int k = 0;
            for (; k + 1 < kend; k += 2) { step(k, in, in2); step(k + 1, in2, in); }
            if (k < kend) step(k, in, in2);
        }
        if (lbT >= 0 && (lbH > bestH || lbT + j < bestR || (lbT + j == bestR && th_ext_rank(bestR, lbT, qlen, tlen) < th_ext_rank(bestR, bestT, qlen, tlen)))) { bestH = lbH; bestR = lbT + j; bestT = lbT; }      // (as th_ksw_ext_wave)
        prs = rs; ppre = re;
        TH_WAVE_FENCE();
    }
    int bestRank = bestT >= 0 ? th_ext_rank(bestR, bestT, qlen, tlen) : 0x7fffffff;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int oH2 = __shfl_xor(bestH, d), oR = __shfl_xor(bestR, d), oK = __shfl_xor(bestRank, d), oT = __shfl_xor(bestT, d);
        const bool take = oH2 > bestH || (oH2 == bestH && (oR < bestR || (oR == bestR && oK < bestRank)));
        if (take) { bestH = oH2; bestR = oR; bestRank = oK; bestT = oT; }
    }
    if (bestH > 0) { *max_t = bestT; *max_q = bestR - bestT; } else { *max_t = -1; *max_q = -1; }
}

// ---- th_ksw_ext_band_wave on packed int16 pairs: TWO columns per lane (round 4) ----
// Same cells, same pruning, same result as th_ksw_ext_band_wave above; strips of 128 columns.  Lane l owns columns j0 + 2l (low half) and
// j0 + 2l + 1 (high half), the high half ONE ROW BEHIND the low one: at step s the low half is on row R - 2l, the high half on row R - 2l - 1
// (R = rs + s), so that the high half's left neighbour is what the low half produced a step ago, and the low half's left neighbour is what
// lane l - 1's high half produced a step ago.  Everything that flows to the right -- H, F' and the target base -- therefore moves by the same
// two instructions per step: a DPP wave shift (lane 0 takes the strip's left border from the inbox, in the high half of the injected word) and
// a v_alignbit that puts (own low half, left lane's high half) into (high, low).  The diagonal input is that left value one step later.
// Rows outside a column's share of the sweep -- rs .. re, the band |i - j| <= W, columns beyond the query -- are two per-lane constants
// (LOB, HIB); -32768 is minus infinity (the caller guarantees qlen + tlen <= TH_KSW_PK_MAXLEN: as in th_ksw_band_ad_wave every real value lies
// within +-(qlen + tlen + 8), and an unreachable in-band cell can drift upwards by one per step from -32768, which stays below all of them).
// Per step ~40 vector instructions for 128 cells where the one-column engine spends ~32 for 64.
#define TH_KSW_PK_MAXLEN 24000
__device__ void th_ksw_ext_band_pk_wave(const uint8_t *__restrict__ q, int qs, int qlen, const uint8_t *__restrict__ t, int ts, int tlen, int W, int slb,
                                        int *__restrict__ bnd, int (*s_in)[TH_WAVE][8], int *max_q, int *max_t)
{
    const int lane = TH_LANE();
    int2 *bnd2 = (int2 *)bnd;                 // {H, F'} of the parked column (ints; minus infinity = -32768 or below)
    int bestH = 0, bestR = 0x7fffffff, bestT = -1;
    int prs = 0, ppre = -1;
    const uint32_t ONE2 = TH_AD_ONE, QQ = TH_PK_DUP(TH_KQ), EE = TH_PK_DUP(TH_KE);
    for (int j0 = 0; j0 < qlen; j0 += 2 * TH_WAVE) {
        const int c0 = j0 + 2 * lane, c1 = c0 + 1;
        const bool ok0 = c0 < qlen, ok1 = c1 < qlen;
        const uint32_t Qp = TH_PK(ok0 ? q[(long long)c0 * qs] : 4, ok1 ? q[(long long)c1 * qs] : 4);
        const uint32_t NQ = th_ad_is4(Qp);
        const bool first = j0 == 0, last = j0 + 2 * TH_WAVE >= qlen;
        int rs = max(0, j0 - W), re = min(tlen - 1, j0 + 2 * TH_WAVE - 1 + W);
        if (j0 > W) {                                                          // entries through the parked column only: the rows its live cells can reach
            int vmin = 0x7fffffff, vmax = -1;
            const int B = qlen - j0;
            for (int r0 = prs; r0 <= ppre; r0 += TH_WAVE) {
                const int row = r0 + lane;
                if (row <= ppre) {
                    const int H = bnd2[row].x, A = tlen - 1 - row;
                    if (H + min(A, B) >= slb) { vmin = min(vmin, row); vmax = max(vmax, row + 2 * TH_WAVE + min(H + B - slb, (H + A - slb) >> 1)); }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { vmin = min(vmin, __shfl_xor(vmin, d)); vmax = max(vmax, __shfl_xor(vmax, d)); }
            rs = max(rs, vmin); re = min(re, vmax);
        }
        rs = TH_UNI(rs); re = TH_UNI(re);
        if (rs > re) break;
        const int nsteps = re - rs + 2 * TH_WAVE;
        // rows of each half's column inside the sweep: [max(rs, c - W), min(re, c + W)]; nothing for a column beyond the query
        const uint32_t LOB = TH_PK(ok0 ? max(rs, c0 - W) : 32767, ok1 ? max(rs, c1 - W) : 32767), HIB = TH_PK(ok0 ? min(re, c0 + W) : -32768, ok1 ? min(re, c1 + W) : -32768);
        const uint32_t ETOP = TH_PK(-(TH_KQ + (c0 + 1) * TH_KE) - TH_KQ - TH_KE, -(TH_KQ + (c1 + 1) * TH_KE) - TH_KQ - TH_KE);        // E entering row 0
        const uint32_t HDTOP = TH_PK(c0 == 0 ? 0 : -(TH_KQ + c0 * TH_KE), -(TH_KQ + c1 * TH_KE));                                  // H(-1, c - 1)
        uint32_t rowv = TH_PK(rs - 2 * lane, rs - 2 * lane - 1);
        uint32_t Hout = TH_PK_NEG, Fout = TH_PK_NEG, Tcur = 0x00040004u, Ecol = TH_PK_NEG, Hdiag = TH_PK_NEG;
        uint32_t lbH = TH_PK_DUP(bestH - 1), lbT = TH_PK_DUP(-1);
        int2 nb = { TH_NEG, TH_NEG }; int nt = 4;
        { const int row = rs + lane; if (row < tlen) { nt = t[(long long)row * ts]; if (!first && row >= prs && row <= ppre) nb = bnd2[row]; } }
        for (int s0 = 0; s0 < nsteps; s0 += TH_WAVE) {
            int (*inbox)[8] = s_in[(s0 >> 6) & 1];
            {
                int4 ib;
                // (the injected word's HIGH half is what lane 0's low half takes)
                if (first) { const int hb = -(TH_KQ + (rs + s0 + lane + 1) * TH_KE); ib.x = (int)((uint32_t)hb << 16); ib.y = (int)((uint32_t)(hb - TH_KQ - TH_KE) << 16); }
                else { ib.x = (int)((uint32_t)max(nb.x, -32768) << 16); ib.y = (int)((uint32_t)max(nb.y, -32768) << 16); }
                ib.z = nt << 16; ib.w = 0;
                *(int4 *)&inbox[lane][0] = ib;
            }
            TH_WAVE_FENCE();
            {
                const int row = rs + s0 + TH_WAVE + lane;
                nt = 4; nb.x = TH_NEG; nb.y = TH_NEG;
                if (row < tlen) { nt = t[(long long)row * ts]; if (!first && row >= prs && row <= ppre) nb = bnd2[row]; }
            }
            const int kend = min(TH_WAVE, nsteps - s0);
            const bool top_ok = rs == 0 && s0 < 3 * TH_WAVE;      // a half meets row 0 within the first 128 steps of a sweep that starts at row 0
            int4 in = *(const int4 *)&inbox[0][0];
            auto step = [&](int k, int4 &cin, int4 &nin) {
                const uint32_t L = th_alignbit16(Hout, (uint32_t)th_shr1_inject((int)Hout, cin.x));
                const uint32_t Fl = th_alignbit16(Fout, (uint32_t)th_shr1_inject((int)Fout, cin.y));
                Tcur = th_alignbit16(Tcur, (uint32_t)th_shr1_inject((int)Tcur, cin.z));
                { const int kn = min(k + 1, TH_WAVE - 1); nin = *(const int4 *)&inbox[kn][0]; }
                if (top_ok) { const uint32_t TOPM = th_pk_negmask_op(th_pk_sub(rowv, ONE2)); Ecol = th_sel(TOPM, ETOP, Ecol); Hdiag = th_sel(TOPM, HDTOP, Hdiag); }      // rows <= 0: the top border's values
                const uint32_t EM = th_pk_negmask_op(th_pk_sub(Tcur ^ Qp, ONE2));
                uint32_t sc = th_sel(EM, ONE2, 0xfffefffeu);
                sc = th_sel(th_ad_is4(Tcur) | NQ, 0xffffffffu, sc);
                const uint32_t h = th_pk_max(th_pk_max(th_pk_add(Hdiag, sc), Ecol), Fl);
                Hdiag = L;
                const uint32_t hq = th_pk_sub(h, QQ);
                const uint32_t nE = th_pk_sub(th_pk_max(Ecol, hq), EE), nF = th_pk_sub(th_pk_max(Fl, hq), EE);
                const uint32_t OUT = th_pk_negmask_op(th_pk_sub(rowv, LOB) | th_pk_sub(HIB, rowv));
                Hout = th_sel(OUT, TH_PK_NEG, h); Ecol = th_sel(OUT, TH_PK_NEG, nE); Fout = th_sel(OUT, TH_PK_NEG, nF);
                const uint32_t upd = th_pk_negmask_op(th_pk_sub(lbH, Hout));                    // a strictly greater H: within a strip a half's anti-diagonal only grows
                lbH = th_pk_max(lbH, Hout); lbT = th_sel(upd, rowv, lbT);
                if (!last && lane == 63) {
                    const int row = rs + s0 + k - 2 * 63 - 1;
                    if (row >= rs && row <= re) { int2 o; o.x = TH_PK_HI(Hout); o.y = TH_PK_HI(Fout); bnd2[row] = o; }
                }
                rowv = th_pk_add(rowv, ONE2);
            };
            int4 in2;
            int k = 0;
            for (; k + 1 < kend; k += 2) { step(k, in, in2); step(k + 1, in2, in); }
            if (k < kend) step(k, in, in2);
        }
        // this strip's two columns into the lane's running best (as th_ksw_ext_wave does per strip)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int lH = hh ? TH_PK_HI(lbH) : TH_PK_LO(lbH), lT = hh ? TH_PK_HI(lbT) : TH_PK_LO(lbT), jc = hh ? c1 : c0;
            if (lT >= 0 && (lH > bestH || (lH == bestH && (lT + jc < bestR || (lT + jc == bestR && th_ext_rank(bestR, lT, qlen, tlen) < th_ext_rank(bestR, bestT, qlen, tlen)))))) { bestH = lH; bestR = lT + jc; bestT = lT; }
        }
        prs = rs; ppre = re;
        TH_WAVE_FENCE();
    }
    int bestRank = bestT >= 0 ? th_ext_rank(bestR, bestT, qlen, tlen) : 0x7fffffff;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int oH2 = __shfl_xor(bestH, d), oR = __shfl_xor(bestR, d), oK = __shfl_xor(bestRank, d), oT = __shfl_xor(bestT, d);
        const bool take = oH2 > bestH || (oH2 == bestH && (oR < bestR || (oR == bestR && oK < bestRank)));
        if (take) { bestH = oH2; bestR = oR; bestRank = oK; bestT = oT; }
    }
    if (bestH > 0) { *max_t = bestT; *max_q = bestR - bestT; } else { *max_t = -1; *max_q = -1; }
}

// ---- th_ksw_global_band_wave on packed int16 pairs: two columns per lane (round 4; geometry as th_ksw_ext_band_pk_wave) ----
// Same band [lo, hi], same pruning with a lower bound slb, same tie rules and path functional (CH / T1 / T2: the '=' columns along ksw_backtrack's
// path), same certificate -- interchangeable with th_ksw_global_band_wave bit for bit.  Five quantities flow to the right (H, F', CH, T2 and the
// target base), two instructions each; comparisons are sign bits of saturating differences, selections are v_bfi.  The caller guarantees
// qlen + tlen <= TH_KSW_PK_MAXLEN ('=' counts are at most min(qlen, tlen)).  bnd: ((tlen + 63) & ~63) * 4 ints; s_in: two 64-row inbox blocks.
// WITH_R: also the back-projection of ksw2_backtrack_left_end for q_left_ext (R0 / R1 / R2 of th_ksw_global_wave: the target bases the path consumes
// from the point where the query's last q_left_ext bases begin), *tback; bnd then holds 6 ints per row.  Certified like iden_n: every optimal path
// and every candidate that ties with one lies inside the band, so the functional is the full matrix's.
template <bool WITH_R>
__device__ int th_ksw_global_band_pk_wave(const uint8_t *__restrict__ q, int qlen, const uint8_t *__restrict__ t, int tlen, int W, int *__restrict__ bnd, int (*s_in)[TH_WAVE][8], int *band_score, int slb = TH_NEG,
                                          int q_left_ext = 0, int *tback = nullptr)
{
    const int lane = TH_LANE();
    const int D = tlen - qlen;
    const int lo = min(0, D) - W, hi = max(0, D) + W;
    int4 *bnd4 = (int4 *)bnd;
    const int tpad = (tlen + 63) & ~63;
    int2 *bndr = (int2 *)(bnd + 4 * tpad);    // {R0, R2} of the parked column (WITH_R)
    const int jstar = qlen - q_left_ext;      // first column of the query suffix (WITH_R)
    const uint32_t ONE2 = TH_AD_ONE, QQ = TH_PK_DUP(TH_KQ), EE = TH_PK_DUP(TH_KE);
    uint32_t res_c = 0, res_h = TH_PK_NEG, res_r = 0;
    int prs = 0, ppre = -1;
    for (int j0 = 0; j0 < qlen; j0 += 2 * TH_WAVE) {
        const int c0 = j0 + 2 * lane, c1 = c0 + 1;
        const bool ok0 = c0 < qlen, ok1 = c1 < qlen;
        const uint32_t Qp = TH_PK(ok0 ? q[c0] : 4, ok1 ? q[c1] : 4);
        const uint32_t NQ = th_ad_is4(Qp);
        const bool first = j0 == 0, last = j0 + 2 * TH_WAVE >= qlen;
        int rs = max(0, j0 + lo), re = min(tlen - 1, j0 + 2 * TH_WAVE - 1 + hi);
        if (slb > TH_NEG && j0 + lo > 0) {
            int vmin = 0x7fffffff, vmax = -1;
            const int b = qlen - j0;
            for (int r0 = prs; r0 <= ppre; r0 += TH_WAVE) {
                const int row = r0 + lane;
                if (row <= ppre) {
                    const int H = bnd4[row].x, a = tlen - 1 - row, dab = a > b ? a - b : b - a;
                    if (H + min(a, b) - dab * TH_KE >= slb) { vmin = min(vmin, row); vmax = max(vmax, row + 2 * TH_WAVE + min(H + b - slb, (H + a - slb) >> 1)); }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { vmin = min(vmin, __shfl_xor(vmin, d)); vmax = max(vmax, __shfl_xor(vmax, d)); }
            rs = max(rs, vmin); re = min(re, vmax);
        }
        rs = TH_UNI(rs); re = TH_UNI(re);
        if (rs > re) { *band_score = TH_NEG; return -1; }
        const int nsteps = re - rs + 2 * TH_WAVE;
        const uint32_t LOB = TH_PK(ok0 ? max(rs, c0 + lo) : 32767, ok1 ? max(rs, c1 + lo) : 32767), HIB = TH_PK(ok0 ? min(re, c0 + hi) : -32768, ok1 ? min(re, c1 + hi) : -32768);
        // the top border (row -1) where it lies inside the band: cell (-1, c) has d = -1 - c, cell (-1, c - 1) has d = -c
        const int et0 = (-1 - c0 >= lo) ? -(TH_KQ + (c0 + 1) * TH_KE) - TH_KQ - TH_KE : -32768, et1 = (-1 - c1 >= lo) ? -(TH_KQ + (c1 + 1) * TH_KE) - TH_KQ - TH_KE : -32768;
        const int hd0 = c0 == 0 ? 0 : ((-c0 >= lo) ? -(TH_KQ + c0 * TH_KE) : -32768), hd1 = (-c1 >= lo) ? -(TH_KQ + c1 * TH_KE) : -32768;
        const uint32_t ETOP = TH_PK(et0, et1), HDTOP = TH_PK(hd0, hd1);
        // the step at which the owner of cell (tlen - 1, qlen - 1) computes it (last strip): row R - 2 l - half = tlen - 1
        const int own_l = ((qlen - 1) - j0) >> 1, own_h = ((qlen - 1) - j0) & 1;
        const int rstar = tlen - 1 + 2 * own_l + own_h;
        uint32_t rowv = TH_PK(rs - 2 * lane, rs - 2 * lane - 1);
        uint32_t Hout = TH_PK_NEG, Fout = TH_PK_NEG, Cout = 0, T2out = 0, Tcur = 0x00040004u, Ecol = TH_PK_NEG, T1 = 0, Hdiag = TH_PK_NEG, Cdiag = 0;
        uint32_t R0out = 0, R2out = 0, R1 = 0, R0diag = 0;
        const uint32_t JSM = (c0 == jstar ? 0xffffu : 0u) | (c1 == jstar ? 0xffff0000u : 0u);       // this lane's column(s) where the query suffix begins
        const uint32_t TL2 = TH_PK_DUP(tlen);
        int4 nb = { TH_NEG, TH_NEG, 0, 0 }; int2 nr = { 0, 0 }; int nt = 4;
        { const int row = rs + lane; if (row < tlen) { nt = t[row]; if (!first && row >= prs && row <= ppre) { nb = bnd4[row]; if (WITH_R) nr = bndr[row]; } } }
        for (int s0 = 0; s0 < nsteps; s0 += TH_WAVE) {
            int (*inbox)[8] = s_in[(s0 >> 6) & 1];
            {
                int4 ib; int ibT = nt << 16;
                if (first) {
                    const int row = rs + s0 + lane;
                    const bool inb = row + 1 <= hi;                     // border column -1 has d = row + 1
                    const int hb = inb ? -(TH_KQ + (row + 1) * TH_KE) : -32768, fb = inb ? hb - TH_KQ - TH_KE : -32768;
                    ib.x = (int)((uint32_t)hb << 16); ib.y = (int)((uint32_t)fb << 16); ib.z = 0; ib.w = 0;
                } else { ib.x = (int)((uint32_t)max(nb.x, -32768) << 16); ib.y = (int)((uint32_t)max(nb.y, -32768) << 16); ib.z = (int)((uint32_t)nb.z << 16); ib.w = (int)((uint32_t)nb.w << 16); }
                *(int4 *)&inbox[lane][0] = ib; inbox[lane][4] = ibT;
                if (WITH_R) { inbox[lane][5] = first ? 0 : (int)((uint32_t)nr.x << 16); inbox[lane][6] = first ? 0 : (int)((uint32_t)nr.y << 16); }
            }
            TH_WAVE_FENCE();
            {
                const int row = rs + s0 + TH_WAVE + lane;
                nt = 4; nb.x = TH_NEG; nb.y = TH_NEG; nb.z = 0; nb.w = 0; nr.x = 0; nr.y = 0;
                if (row < tlen) { nt = t[row]; if (!first && row >= prs && row <= ppre) { nb = bnd4[row]; if (WITH_R) nr = bndr[row]; } }
            }
            const int kend = min(TH_WAVE, nsteps - s0);
            const bool top_ok = rs == 0 && s0 < 3 * TH_WAVE;
            int4 in = *(const int4 *)&inbox[0][0]; int inT = inbox[0][4];
            auto step = [&](int k, int4 &cin, int &cinT, int4 &nin, int &ninT) {
                uint32_t R0l = 0, R2l = 0;
                if (WITH_R) {       // (the border words of R0 / R2 are read where they are used: two more broadcast reads of the step's inbox entry)
                    R0l = th_alignbit16(R0out, (uint32_t)th_shr1_inject((int)R0out, inbox[k][5]));
                    R2l = th_alignbit16(R2out, (uint32_t)th_shr1_inject((int)R2out, inbox[k][6]));
                }
                const uint32_t L = th_alignbit16(Hout, (uint32_t)th_shr1_inject((int)Hout, cin.x));
                const uint32_t Fl = th_alignbit16(Fout, (uint32_t)th_shr1_inject((int)Fout, cin.y));
                const uint32_t Cl = th_alignbit16(Cout, (uint32_t)th_shr1_inject((int)Cout, cin.z));
                const uint32_t T2l = th_alignbit16(T2out, (uint32_t)th_shr1_inject((int)T2out, cin.w));
                Tcur = th_alignbit16(Tcur, (uint32_t)th_shr1_inject((int)Tcur, cinT));
                { const int kn = min(k + 1, TH_WAVE - 1); nin = *(const int4 *)&inbox[kn][0]; ninT = inbox[kn][4]; }
                if (top_ok) {
                    const uint32_t TOPM = th_pk_negmask_op(th_pk_sub(rowv, ONE2));      // rows <= 0
                    Ecol = th_sel(TOPM, ETOP, Ecol); T1 &= ~TOPM; Hdiag = th_sel(TOPM, HDTOP, Hdiag); Cdiag &= ~TOPM;
                    if (WITH_R) { R1 = th_sel(TOPM, TL2, R1); R0diag = th_sel(TOPM, TL2, R0diag); }      // a path that leaves through the top border still owes an insertion: all of t is consumed
                }
                const uint32_t EM = th_pk_negmask_op(th_pk_sub(Tcur ^ Qp, ONE2));
                uint32_t sc = th_sel(EM, ONE2, 0xfffefffeu);
                sc = th_sel(th_ad_is4(Tcur) | NQ, 0xffffffffu, sc);
                // ties prefer the diagonal over E over F (ksw2_extz2_sse.c:177-195)
                const uint32_t diag = th_pk_add(Hdiag, sc);
                const uint32_t g_e = th_pk_negmask_op(th_pk_sub(diag, Ecol));                 // E > diagonal
                const uint32_t h1 = th_pk_max(diag, Ecol);
                const uint32_t g_f = th_pk_negmask_op(th_pk_sub(h1, Fl));                     // F > both
                const uint32_t h = th_pk_max(h1, Fl);
                const uint32_t hq = th_pk_sub(h, QQ);
                const uint32_t xe = th_pk_negmask_op(th_pk_sub(hq, Ecol)), xf = th_pk_negmask_op(th_pk_sub(hq, Fl));
                const uint32_t ch = th_sel(g_f, T2l, th_sel(g_e, T1, th_pk_sub(Cdiag, EM)));
                const uint32_t t1 = th_sel(xe, T1, ch), t2 = th_sel(xf, T2l, ch);
                uint32_t r0 = 0;
                if (WITH_R) {
                    // in the suffix's first column the counts start: tlen - i on the diagonal, tlen - 1 - i out of a horizontal gap (th_ksw_global_wave)
                    const uint32_t TLr = th_pk_sub(TL2, rowv);
                    const uint32_t A_ = th_sel(JSM, TLr, R0diag), B_ = th_sel(JSM, th_pk_sub(TLr, ONE2), R2l);
                    r0 = th_sel(g_f, B_, th_sel(g_e, R1, A_));
                    const uint32_t r1 = th_sel(xe, R1, r0), r2 = th_sel(xf, B_, r0);
                    R0diag = R0l; R1 = r1; R0out = r0; R2out = r2;
                }
                Hdiag = L; Cdiag = Cl; T1 = t1;
                const uint32_t nE = th_pk_sub(th_pk_max(Ecol, hq), EE), nF = th_pk_sub(th_pk_max(Fl, hq), EE);
                const uint32_t OUT = th_pk_negmask_op(th_pk_sub(rowv, LOB) | th_pk_sub(HIB, rowv));
                Hout = th_sel(OUT, TH_PK_NEG, h); Ecol = th_sel(OUT, TH_PK_NEG, nE); Fout = th_sel(OUT, TH_PK_NEG, nF); Cout = ch; T2out = t2;
                if (last && rs + s0 + k == rstar) { res_c = ch; res_h = h; res_r = r0; }          // (wave-uniform test; the owner's lane and half are picked below)
                if (!last && lane == 63) {
                    const int row = rs + s0 + k - 2 * 63 - 1;
                    if (row >= rs && row <= re) {
                        int4 o; o.x = TH_PK_HI(Hout); o.y = TH_PK_HI(Fout); o.z = TH_PK_HI(Cout); o.w = TH_PK_HI(T2out); bnd4[row] = o;
                        if (WITH_R) { int2 o2; o2.x = TH_PK_HI(R0out); o2.y = TH_PK_HI(R2out); bndr[row] = o2; }
                    }
                }
                rowv = th_pk_add(rowv, ONE2);
            };
            int4 in2; int in2T;
            int k = 0;
            for (; k + 1 < kend; k += 2) { step(k, in, inT, in2, in2T); step(k + 1, in2, in2T, in, inT); }
            if (k < kend) step(k, in, inT, in2, in2T);
        }
        if (last) {
            const uint32_t rc = (uint32_t)TH_READLANE((int)res_c, own_l), rh = (uint32_t)TH_READLANE((int)res_h, own_l);
            res_c = (uint32_t)(own_h ? TH_PK_HI(rc) : TH_PK_LO(rc)); res_h = (uint32_t)(own_h ? TH_PK_HI(rh) : TH_PK_LO(rh));
            if (WITH_R) { const uint32_t rr = (uint32_t)TH_READLANE((int)res_r, own_l); res_r = (uint32_t)(own_h ? TH_PK_HI(rr) : TH_PK_LO(rr)); }
        }
        prs = rs; ppre = re;
        TH_WAVE_FENCE();
    }
    const int iden = (int)res_c, hsc = (int)res_h <= -32768 ? TH_NEG : (int)res_h;
    const int gmin = 2 * W + 2 + (D < 0 ? -D : D);
    *band_score = hsc;
    if (WITH_R && tback) *tback = (int)res_r;
    if (2LL * hsc > (long long)(qlen + tlen - gmin) - 4 * TH_KQ - 2LL * gmin * TH_KE) return iden;
    return -1;
}

// Global alignment with iden_n and the back-projection, cheapest exact engine first (the partition's bridging alignments, th_k_partition.hip):
// a certified band of half-width tlen / 10 + 24 on the packed strip engine, then the band its banded score names (pruned with that score), then
// the full matrix.  s_in: the strip engines' inbox (two 64-row blocks); bnd: ((tlen + 63) & ~63) * 6 ints.  Same results as th_ksw_global_wave<true>.
__device__ __forceinline__ int th_ksw_global_r_wave(const uint8_t *__restrict__ q, int qlen, const uint8_t *__restrict__ t, int tlen, int q_left_ext, int *__restrict__ bnd, int (*s_in)[TH_WAVE][8], int *tback, bool banded)
{
    if (banded && qlen + tlen <= TH_KSW_PK_MAXLEN && qlen >= 2 * TH_WAVE) {
        const int D = tlen - qlen, aD = D < 0 ? -D : D;
        const int W = tlen / 10 + 24;
        if (64 + 2 * W + aD < (tlen * 3) / 4) {                       // (a band that is most of the matrix saves nothing)
            int s1 = TH_NEG, tb = 0;
            int v = th_ksw_global_band_pk_wave<true>(q, qlen, t, tlen, W, bnd, s_in, &s1, TH_NEG, q_left_ext, &tb);
            if (v < 0 && s1 > TH_NEG) {
                const long long g2 = ((long long)qlen + tlen - 2LL * s1 - 4 * TH_KQ) / (1 + 2 * TH_KE) + 2;
                const long long W2 = (g2 - 2 - aD + 1) / 2 + 8;
                if (W2 > W && W2 < 30000) { const int lb = s1; v = th_ksw_global_band_pk_wave<true>(q, qlen, t, tlen, (int)W2, bnd, s_in, &s1, lb, q_left_ext, &tb); }
            }
            if (v >= 0) { *tback = tb; return v; }
        }
    }
    return th_ksw_global_wave<true>(q, qlen, t, tlen, q_left_ext, bnd, tback);
}
