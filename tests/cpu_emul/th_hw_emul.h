// tests/cpu_emul/th_hw_emul.h -- TEST-ONLY stand-in for tidehunter_amd/csrc/th_hw.h: the kernels' hardware primitives restated over
// the fiber emulation of hip_emul.h (rendezvous shuffles instead of DPP, plain C for the packed int16 instructions).  Pre-included
// by tests/cpu_emul/Makefile; defines TH_HW_H so that the product header is skipped.  Never part of libtidehunter_amd.so.
#ifndef TH_HW_H
#define TH_HW_H
#include "th_rt_emul.h"
// (token-identical to th_dev.h, which defines them after this header has been read)
#define TH_WAVE 64
#define TH_PK_NEG 0x80008000u            // (-32768, -32768): "minus infinity" in both halves
#define TH_PK_LO(x) ((int)(short)((x) & 0xffffu))
#define TH_PK_HI(x) ((int)(short)((uint32_t)(x) >> 16))
#define TH_PK(lo, hi) (((uint32_t)(lo) & 0xffffu) | ((uint32_t)(hi) << 16))
#define TH_PK_DUP(v) TH_PK((v), (v))
#define TH_WAVE_FENCE() do { (void)hipemul::wave_exchange(0); } while (0)

#define TH_READLANE(v, l) __shfl((v), (l))
#define TH_WRITELANE(old, v, l) (TH_LANE() == (l) ? (v) : (old))
#define th_writelane6(a, b, c, d, e, f, va, vb, vc, vd, ve, vf, l) do { if (TH_LANE() == (l)) { a = (va); b = (vb); c = (vc); d = (vd); e = (ve); f = (vf); } } while (0)
__device__ __forceinline__ int th_shr1(int v, int fill) { int o = __shfl_up(v, 1); return TH_LANE() == 0 ? fill : o; }
__device__ __forceinline__ int th_shl1(int v, int fill) { int o = __shfl_down(v, 1); return TH_LANE() == 63 ? fill : o; }
__device__ __forceinline__ int th_shr1_z(int v) { return th_shr1(v, 0); }
__device__ __forceinline__ int th_shl1_z(int v) { return th_shl1(v, 0); }
// lane l <- lane l-1, lane 0 <- lane 0 of `inj` (a per-lane register: only its lane 0 matters)
__device__ __forceinline__ int th_shr1_inject(int v, int inj) { int o = __shfl_up(v, 1); return TH_LANE() == 0 ? inj : o; }
// rotate towards lane 0: lane l <- lane l+1, lane 63 <- lane 0
__device__ __forceinline__ int th_rol1(int v) { return __shfl(v, (TH_LANE() + 1) & 63); }
__device__ __forceinline__ int th_wave_incl_max(int v)
{
    const int lane = TH_LANE();
    for (int d = 1; d < TH_WAVE; d <<= 1) { int o = __shfl_up(v, d); if (lane >= d) v = max(v, o); }
    return v;
}
__device__ __forceinline__ int th_wave_incl_sum(int v)
{
    const int lane = TH_LANE();
    for (int d = 1; d < TH_WAVE; d <<= 1) { int o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}

#define TH_LOAD_L2(p) (*(p))

#define TH_VM_WAIT0() ((void)0)
#define TH_WAVES_PER_SIMD(n)

#define TH_UNI(x) (x)

#define TH_UNI_PTR(T, p) (p)
#define TH_OPAQUE_V(x) ((void)0)
#define TH_KEEP_V(x) ((void)0)
#define TH_OPAQUE_S(x) ((void)0)
#define TH_NOINLINE
#define TH_LDS_SLACK_INDEX(i, cap) (min((int)(i), (int)(cap)))       // (a static array here: stay inside it)

#define TH_BALLOT(b) __ballot(b)
#define TH_STORE_U32_SADDR(base, off, v) (*(uint32_t *)((char *)(base) + (uint32_t)(off)) = (v))
#define TH_STORE_U16_SADDR(base, off, v) (*(uint16_t *)((char *)(base) + (uint32_t)(off)) = (uint16_t)(v))
#define TH_LDS_READ64_ASYNC(p) (*(p))
#define TH_LDS_ASYNC_WAIT(v) ((void)0)
#define TH_LDS_ASYNC_TAKE(dst, v) ((dst) = (v))
#define TH_LANE_OF_MASK(m, lane) (((m) >> (lane)) & 1ull)
#define TH_CLZ_NZ(x) ((unsigned)(x) ? __builtin_clz((unsigned)(x)) : 32)
static inline int __mul24(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

static inline int th_sat16_(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
static inline uint32_t th_pk_add(uint32_t a, uint32_t b) { return TH_PK(th_sat16_(TH_PK_LO(a) + TH_PK_LO(b)), th_sat16_(TH_PK_HI(a) + TH_PK_HI(b))); }
static inline uint32_t th_pk_sub(uint32_t a, uint32_t b) { return TH_PK(th_sat16_(TH_PK_LO(a) - TH_PK_LO(b)), th_sat16_(TH_PK_HI(a) - TH_PK_HI(b))); }
static inline uint32_t th_pk_max(uint32_t a, uint32_t b) { return TH_PK(TH_PK_LO(a) > TH_PK_LO(b) ? TH_PK_LO(a) : TH_PK_LO(b), TH_PK_HI(a) > TH_PK_HI(b) ? TH_PK_HI(a) : TH_PK_HI(b)); }
static inline uint32_t th_pk_negmask(uint32_t a) { return (TH_PK_LO(a) < 0 ? 0xffffu : 0u) | (TH_PK_HI(a) < 0 ? 0xffff0000u : 0u); }
static inline uint32_t th_alignbit16(uint32_t hi, uint32_t lo) { return (lo >> 16) | (hi << 16); }   // (lo.hi, hi.lo)
#define TH_PK_SUB_IMM(x, K) th_pk_sub((x), TH_PK_DUP(K))
static inline int th_smax(int a, int b) { return a > b ? a : b; }
static inline int th_smin(int a, int b) { return a < b ? a : b; }
static inline int th_smax0(int a) { return a > 0 ? a : 0; }
static inline uint32_t th_pk_neg_outside(uint32_t m) { return ~m & 0x80008000u; }
static inline uint32_t th_pk_score_sel(uint32_t qx) { return (((qx << 12) | qx) & 0x000f000fu) | 0x0c000c00u; }
static inline uint32_t th_pk_negmask_op(uint32_t a) { return th_pk_negmask(a); }
static inline uint32_t th_pk_sign(uint32_t a) { return (TH_PK_LO(a) < 0 ? 1u : 0u) | (TH_PK_HI(a) < 0 ? 0x10000u : 0u); }
static inline uint32_t th_ubfe(uint32_t v, uint32_t off, uint32_t w) { return (v >> (off & 31)) & ((1u << w) - 1); }
static inline uint32_t th_cvt_pk_i16(int lo, int hi) { return TH_PK(th_sat16_(lo), th_sat16_(hi)); }     // two ints to a saturated int16 pair
// v_perm_b32: result byte k is picked by selector byte k: 0..3 byte of s1, 4..7 byte of s0, 12 the constant 0 (the other codes are not used here)
static inline int th_mbcnt(unsigned long long m) { return __builtin_popcountll(m & ((1ull << TH_LANE()) - 1)); }
#define TH_ALN_OCC              // (th_k_align.hip: an occupancy attribute of the device build)
#define TH_PERM_SIGNS 0x0b090a08u
#define TH_PERM_SIGNS8 0x09080b0au
static inline uint32_t th_and_or(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }
static inline uint32_t th_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const unsigned long long src = ((unsigned long long)s0 << 32) | s1; uint32_t r = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t c = (sel >> (8 * k)) & 0xff;
        // 0..7: a byte of {s0, s1}; 8..11: 0xff / 0x00 from the sign of byte 1, 3, 5, 7; 12: 0x00; 13 and above: 0xff
        r |= (c < 8 ? (uint32_t)((src >> (8 * c)) & 0xff) : c < 12 ? (((src >> (16 * (c - 8) + 15)) & 1) ? 0xffu : 0u) : c >= 13 ? 0xffu : 0u) << (8 * k);
    }
    return r;
}
__device__ __forceinline__ uint32_t th_wave_incl_pkmax(uint32_t v)
{
    const int lane = TH_LANE();
    for (int d = 1; d < TH_WAVE; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, d); if (lane >= d) v = th_pk_max(v, o); }
    return v;
}

#endif
