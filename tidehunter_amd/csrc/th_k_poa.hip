// th_k_poa.hip -- stage 4: partial-order-alignment consensus over the repeat units of one run (gfx950).
//
// Stands where the reference calls abpoa_msa() (src/abpoa_cons.c:30-120; abPOA itself is an absent
// submodule -- see DESIGN.md "POA parity").  The algorithm is the one oracle/tho_poa.c states:
// progressive alignment of units 1..n-1 onto a graph seeded by unit 0, adaptive band
// (w = 10 + qlen/100 around the row maxima of the predecessor rows and the expected diagonal),
// convex gap cost (two affine pieces), row-column consensus by most frequent symbol, graph kept
// in "insertion order".
//
// One consensus job per wavefront, waves loop over the job list (persistent launch).  Every
// phase is wave-parallel except the traceback walk:
//   remain     : longest path to the sink by 64-node chunks, in-chunk dependencies resolved by
//                register relaxation (no memory round trips);
//   row tables : per row (order index) base, degrees, predecessor / successor ROW indices, so the DP
//                loop never chases node ids;
//   DP         : rows whose only predecessor is the cached previous row run two columns per lane on packed int16
//                pairs, other rows one column per lane; F1/F2 are max-plus prefix scans (DPP); the three previous
//                rows live in an LDS int16 ring, so the common row has no dependent HBM load; only the 4-byte
//                direction word of a cell goes to HBM unless a far row needs {H,E1',E2'};
//   traceback  : sequential, but reads its direction words from 16-row x 4-diagonal tiles loaded by the whole
//                wave, and emits plain diagonal runs of a tile column in one step;
//   merge      : one alignment column per lane (node choice, ring splice, new ids by prefix sum,
//                edges: every node is `from` once and `to` once per path, so lanes never collide);
//   order      : positions of new nodes by a prefix sum over the old order.
#include "th_dev.h"

#define TH_POA_BLOCK 64
#define TH_POA_WB 10
#ifndef TH_POA_QCAP
#define TH_POA_QCAP 3072        // query bases staged in LDS by the int16-ring variant
#endif
#define TH_POA_QCAP_WIDE 8192   // ... by the 32-bit-ring variant (longer units read the query from HBM: QCAP = 0)

// Direction words.  In registers (traceback) a word has the layout of the TH_D_* macros below.  In HBM a cell of a row with at
// most four predecessors -- every packed row, nearly every other row -- is a 16-bit word made of raw comparison results, laid out
// as the packed row's byte permutes deliver them (v_perm_b32 with the sign selectors turns the sign bits of TWO packed differences
// into four bytes of 0xff / 0x00: the upper byte of a cell's word takes the first difference of a pair, the lower byte the second):
//   bits 8, 0, 9, 1    "strictly better than everything before it" for E1, E2, F1, F2 in that order (the source is the last one set)
//   bits 10, 2, 11, 3  XE1, XE2, XF1, XF2: the gap was extended rather than opened
//   bits 13..12 / 15..14 / 5..4  edge index of the predecessor that gave the diagonal / the E1 / the E2 value
// TH_POA_D8 = 1: a row with exactly ONE predecessor (three rows in four at the headline shape) needs no edge indices: its cells are 8-BIT words --
// bits 4, 0, 5, 1 for E1, E2, F1, F2, bits 6, 2, 7, 3 for XE1, XE2, XF1, XF2: the low nibble of either byte of the 16-bit word -- which the
// packed row assembles from the same four permutes with the sign selectors in source order (TH_PERM_SIGNS8) and one shift-or, and
// stores two per lane as one short.
// so the packed row spends one instruction per comparison (a permute and an and-or per pair; round 2: two per comparison) instead of
// assembling a 32-bit word per cell, and the traffic of the direction matrix halves.  Rows with more than four predecessors
// store whole cells (int4, the 32-bit word in .w); th_poa_dir() returns the register layout for either kind.
#ifndef TH_POA_D8
#define TH_POA_D8 0             // 1: 8-bit direction words for rows with one predecessor (below).  Measured on the GPU in round 4 at the headline shape and
                                // NOT the default: it takes 11 % off the kernel's HBM traffic (write 383 -> 320 GB per 100 000 reads: the rows are not line-aligned, so
                                // the partial sectors at either end of a 65-byte row are counted whole) and 0.35 MB off a wave's scratch, and costs 13 ms of 459 (the fold of
                                // the two halves, a short store per trip, 16 bytes more register scratch).  The CPU emulation's second build (libth_emul_rel.so) keeps it compiled and tested.
#endif
#define TH_D16_WIDE_DEG 4
#define TH_D16_PM_SHIFT 12
#define TH_D16_PE1_SHIFT 14
#define TH_D16_PE2_SHIFT 4
__device__ __forceinline__ uint32_t th_d16_expand(uint32_t w)
{
    const uint32_t src = (w & 0x002u) ? 4u : (w & 0x200u) ? 3u : (w & 0x001u) ? 2u : (w & 0x100u) ? 1u : 0u;
    return src | ((w >> TH_D16_PM_SHIFT) & 3u) << 3 | ((w >> TH_D16_PE1_SHIFT) & 3u) << 11 | ((w >> TH_D16_PE2_SHIFT) & 3u) << 19
         | ((w >> 10) & 1u) << 27 | ((w >> 2) & 1u) << 28 | ((w >> 11) & 1u) << 29 | ((w >> 3) & 1u) << 30;
}
__device__ __forceinline__ uint32_t th_d8_expand(uint32_t b) { return th_d16_expand((b & 0x0fu) | (b & 0xf0u) << 4); }
__device__ __forceinline__ uint32_t th_d8_from16(uint32_t w) { return (w & 0x0fu) | ((w >> 4) & 0xf0u); }      // (a 16-bit word without edge indices)
// 2-byte units a row of `span` cells (span even) takes among the direction words: 8-bit rows are rounded up to a whole dword
__device__ __forceinline__ int th_d_units(int span, bool d8) { return (TH_POA_D8 && d8) ? ((span >> 1) + 1) & ~1 : span; }
// the 16-bit word from a source (0..4), the four extension flags and the three edge indices (the one-column-per-lane rows)
__device__ __forceinline__ uint32_t th_d16_pack(int src, uint32_t xe1, uint32_t xe2, uint32_t xf1, uint32_t xf2, uint32_t pm, uint32_t pe1, uint32_t pe2)
{
    const uint32_t s = src == 1 ? 0x100u : src == 2 ? 0x001u : src == 3 ? 0x200u : src == 4 ? 0x002u : 0u;
    return s | xe1 << 10 | xe2 << 2 | xf1 << 11 | xf2 << 3 | pm << TH_D16_PM_SHIFT | pe1 << TH_D16_PE1_SHIFT | pe2 << TH_D16_PE2_SHIFT;
}
#define TH_D_SRC(d)  ((d) & 7)
#define TH_D_PM(d)   (((d) >> 3) & 0xff)
#define TH_D_PE1(d)  (((d) >> 11) & 0xff)
#define TH_D_PE2(d)  (((d) >> 19) & 0xff)
#define TH_D_XE1 (1u << 27)
#define TH_D_XE2 (1u << 28)
#define TH_D_XF1 (1u << 29)
#define TH_D_XF2 (1u << 30)

// Consensus column rule when the gap is exactly as frequent as the best base: 0 = the base stays (a later symbol of the scan
// A,C,G,T,N,gap must be strictly more frequent), 1 = the column is dropped.  libabpoa's rule is unverifiable here (absent
// submodule); this constant and THO_POA_GAP_WINS_TIE (oracle/tho.h) are the one line each to change when it can be read.
#define TH_POA_GAP_WINS_TIE 0
#define TH_POA_INL 4
#define TH_POA_RING 4      // recent rows kept in LDS (a row can read the three before it)
#define TH_POA_PAD 4       // minus-infinity shorts in front of every ring row (columns left of the band; clamp target)
#define TH_BIG 0x3fffffff

// Capacities of one job's graph arrays inside its wave's scratch (th_poa_carve), the same on the host (th_poa_need_ints) and in the
// kernel.  `cap` bounds the NODES (= rows): the worst case is one node per base of every unit (sum_len + 2, what rounds 1-3 carved: 39
// arrays of it, 1.75 MB of a headline job's 3.8 MB), the usual case is the longest unit plus ~9 % per further unit; the arrays are sized
// for `factor` times that estimate (1.5 on the first attempt), a job whose graph outgrows them is given back with status 1 and rerun with
// twice the factor -- up to the worst case, which every no-band rerun uses.  fexp: factor = 1.5 * 2^fexp.
struct ThPoaCaps { long long cap, ocap, opn; };
__host__ __device__ static inline ThPoaCaps th_poa_caps(int n_units, int sum_len, int max_len, int fexp, bool full)
{
    ThPoaCaps c;
    const long long worst = (long long)sum_len + 2;
    long long est = (long long)((max_len * (1.0 + 0.09 * (n_units - 1)) + 64) * 1.5) << (fexp < 0 ? 0 : fexp > 20 ? 20 : fexp);
#ifdef TH_POA_NOCAPS
    full = true;
#endif
    c.cap = (full || est > worst) ? worst : est;
    c.ocap = 2LL * (sum_len + n_units + 8);                 // overflow edges (beyond four in / out per node): by edges, not by nodes -- 2 small arrays
    c.opn = c.cap + max_len + 2;                            // alignment operations: a path visits every row and every query base at most once
    return c;
}

// The kernels are compiled in two configurations (th_k_poa_body.h, once per namespace):
//   thpoa_n  ring rows of up to 192 columns (two trips of 128), absolute int16 scores: units of up to ~1500 bases (TH_POA_QCAP staged
//            bases; longer ones whose scores may leave int16 are given back with status 7), 6 waves per SIMD -- and the 32-bit-ring
//            variants <8192, int> / <0, int> for everything else;
//   thpoa_l  ring rows of up to 382 columns (three trips), int16 scores RELATIVE to a running base that follows the row maxima: units
//            of up to TH_POA_QCAP_LONG bases (configs 4 and 5: periods of 2000 and 5000) on the packed path instead of one column per
//            lane in 32 bits, 3 waves per SIMD.
#ifndef TH_POA_OCC
#define TH_POA_OCC 6            // waves per SIMD the thpoa_n kernels are compiled for
#endif
#ifndef TH_POA_RW_N
#define TH_POA_RW_N 192         // widest cached band, thpoa_n
#endif
#define TH_POA_QCAP_LONG 6144   // query bases staged in LDS by thpoa_l
#ifndef TH_POA_RW_L
#define TH_POA_RW_L 382         // widest cached band, thpoa_l (a band of RW columns may start at an odd column: RW + 2 <= 128 x trips)
#endif
#ifndef TH_POA_TRIPS_L
#define TH_POA_TRIPS_L 3
#endif
#ifndef TH_POA_REL_THR
#define TH_POA_REL_THR (-24000) // relative scores below this are minus infinity
#endif
#ifndef TH_POA_REL_HI
#define TH_POA_REL_HI 8000      // a row maximum beyond +- this rebases the ring
#endif
#define TH_POA_RS (TH_POA_PAD + TH_POA_RW + 4)   // shorts per ring array: pad + band (from its even base) + the odd-edge pair

#define TH_POA_NS thpoa_n
#define TH_POA_RW TH_POA_RW_N
#define TH_POA_TRIPS 2
#define TH_POA_REL 0
#define TH_POA_OCC_NS TH_POA_OCC
#include "th_k_poa_body.h"
#undef TH_POA_NS
#undef TH_POA_RW
#undef TH_POA_TRIPS
#undef TH_POA_REL
#undef TH_POA_OCC_NS

#define TH_POA_NS thpoa_l
#define TH_POA_RW TH_POA_RW_L
#define TH_POA_TRIPS TH_POA_TRIPS_L
#define TH_POA_REL 1
#define TH_POA_OCC_NS 3
#include "th_k_poa_body.h"
#undef TH_POA_NS
#undef TH_POA_RW
#undef TH_POA_TRIPS
#undef TH_POA_REL
#undef TH_POA_OCC_NS
#define TH_POA_RW TH_POA_RW_N       // (host-side estimates: th_poa_need_ints)
