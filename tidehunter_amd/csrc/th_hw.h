// th_hw.h -- gfx950 primitives the kernels are written in: wave-scope fences, lane reads / writes (v_readlane, v_writelane via m0),
// lane shifts and scans as DPP moves, packed int16 arithmetic (v_pk_*_i16, v_perm_b32, v_bfi_b32), counter-aware waits.
// Included by th_dev.h only.  (The test-only CPU build of the kernel sources pre-includes tests/cpu_emul/th_hw_emul.h, which defines
// TH_HW_H and provides the same names over its fiber emulation, so that no product header carries test code.)
#ifndef TH_HW_H
#define TH_HW_H
#if defined(TH_SAFE_FENCE)
#define TH_WAVE_FENCE() __threadfence_block()
#else
// Every team in these kernels is ONE wavefront, so the hand-offs are lane-to-lane inside a wave: the
// wave's vector-memory instructions reach its CU's L1 in issue order, which already orders a lane's
// store before another lane's later load.  A wavefront-scope fence keeps the compiler from reordering
// and, unlike __threadfence_block(), does not stall on s_waitcnt vmcnt(0) (the store round trip to L2,
// ~1-2 us) -- that wait per POA row was the dominant cost of the first version.
#define TH_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
#endif

#define TH_READLANE(v, l) __builtin_amdgcn_readlane((v), (l))
// lane l of `old` replaced by the wave-uniform v (v_writelane_b32: value and lane select are scalar operands; gfx9 reads one
// SGPR per vector instruction, so the lane select travels in m0, which is saved and restored around the instruction)
__device__ __forceinline__ int th_writelane(int old, int v, int l)
{
    int keep;
    asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(old), "=&s"(keep) : "s"(v), "s"(l));
    return old;
}
// six registers, one lane select: m0 is loaded once (the compiler is told it is clobbered instead of saving it around each write)
__device__ __forceinline__ void th_writelane6(int &a, int &b, int &c, int &d, int &e, int &f, int va, int vb, int vc, int vd, int ve, int vf, int l)
{
    asm("s_mov_b32 m0, %12\n\tv_writelane_b32 %0, %6, m0\n\tv_writelane_b32 %1, %7, m0\n\tv_writelane_b32 %2, %8, m0\n\t"
        "v_writelane_b32 %3, %9, m0\n\tv_writelane_b32 %4, %10, m0\n\tv_writelane_b32 %5, %11, m0"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(va), "s"(vb), "s"(vc), "s"(vd), "s"(ve), "s"(vf), "s"(l) : "m0");
}
#define TH_WRITELANE(old, v, l) th_writelane((old), (v), (l))
#define TH_DPP(old, src, ctrl, rmask, bmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), (bmask), false)
__device__ __forceinline__ int th_shr1(int v, int fill) { return TH_DPP(fill, v, 0x138, 0xf, 0xf); }   // wave_shr:1
__device__ __forceinline__ int th_shl1(int v, int fill) { return TH_DPP(fill, v, 0x130, 0xf, 0xf); }   // wave_shl:1
// the same with ZERO shifted in: bound_ctrl supplies it, no register has to be filled first (one instruction instead of two)
__device__ __forceinline__ int th_shr1_z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int th_shl1_z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ int th_shr1_inject(int v, int inj) { return TH_DPP(inj, v, 0x138, 0xf, 0xf); }  // lane 0 keeps inj[0]
__device__ __forceinline__ int th_rol1(int v) { return TH_DPP(v, v, 0x134, 0xf, 0xf); }                   // wave_rol:1
// Inclusive scans: four dependent steps inside each row of 16 lanes (row_shr 1, 2, 4, 8: a lane without a source keeps its
// value), then the two row broadcasts -- six vector instructions.  (A variant that shortens the dependency chain with three
// independent shifts and a max3 needs nine plus identity fills; these kernels are bound by instruction issue, not by the chain.)
__device__ __forceinline__ int th_wave_incl_max(int v)
{
    const int id = -0x7fffffff - 1;
    int x = v;
    x = max(x, TH_DPP(id, x, 0x111, 0xf, 0xf));     // row_shr:1
    x = max(x, TH_DPP(id, x, 0x112, 0xf, 0xf));     // row_shr:2
    x = max(x, TH_DPP(id, x, 0x114, 0xf, 0xf));     // row_shr:4
    x = max(x, TH_DPP(id, x, 0x118, 0xf, 0xf));     // row_shr:8
    x = max(x, TH_DPP(id, x, 0x142, 0xa, 0xf));     // row_bcast:15 into rows 1,3
    x = max(x, TH_DPP(id, x, 0x143, 0xc, 0xf));     // row_bcast:31 into rows 2,3
    return x;
}
__device__ __forceinline__ int th_wave_incl_sum(int v)
{
    int x = v;
    x += TH_DPP(0, x, 0x111, 0xf, 0xf);
    x += TH_DPP(0, x, 0x112, 0xf, 0xf);
    x += TH_DPP(0, x, 0x114, 0xf, 0xf);
    x += TH_DPP(0, x, 0x118, 0xf, 0xf);
    x += TH_DPP(0, x, 0x142, 0xa, 0xf);
    x += TH_DPP(0, x, 0x143, 0xc, 0xf);
    return x;
}

// Plain load of a word that was updated with device-scope atomics earlier in the same kernel: atomics execute
// in L2 and do not refresh the CU's vector L1, so the read-back must bypass L1 (global_load ... sc1).
#define TH_LOAD_L2(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// Explicit "all my vector-memory loads have landed".  gfx9 counts loads AND stores in vmcnt, and the compiler places its
// wait at the first use of a loaded register; when that use sits in a loop whose body also issues fire-and-forget
// stores, the wait it inserts there drains the stores of the previous trip on every trip.  Waiting right after the
// (rare) loads instead leaves the steady-state trips without any vmcnt wait.
#define TH_VM_WAIT0() __builtin_amdgcn_s_waitcnt(0x0F70)      // vmcnt(0), expcnt and lgkmcnt untouched
#define TH_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))   // caps the VGPR budget so that n waves fit a SIMD

// A value that is the same in every lane but that the compiler cannot prove uniform (it came from a per-lane load or an LDS
// read): v_readfirstlane moves it to an SGPR, so that what depends on it is scalar code and scalar branches instead of
// vector code under execution masks.
// INVARIANT: the value MUST be the same in every active lane.  The build leaves wave-uniform regions unstructured (-structurizecfg-skip-uniform-regions, csrc/Makefile),
// so a branch on a TH_UNI() value is a SCALAR branch taken by the whole wave: were the value lane-dependent, the lanes that disagree with lane 0 would silently follow lane 0's
// path (with the structurizer they would merely compute lane 0's value).  tests/test_gpu_scale.py::test_gpu_structurizer_flag_changes_no_result compares the two builds.
#define TH_UNI(x) __builtin_amdgcn_readfirstlane(x)

// max / min of wave-uniform values ON THE SCALAR UNIT: for max(max(0, x), y) of three scalars hipcc picks v_mov + v_max3_i32 + v_readfirstlane (and a vector
// compare for the test that follows): four vector instructions per POA row where two scalar ones do
__device__ __forceinline__ int th_smax(int a, int b) { int r; asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ int th_smin(int a, int b) { int r; asm("s_min_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ int th_smax0(int a) { int r; asm("s_max_i32 %0, %1, 0" : "=s"(r) : "s"(a) : "scc"); return r; }
#define TH_UNI_PTR(T, p) ((T *)(((unsigned long long)(uint32_t)TH_UNI((int)((unsigned long long)(p) >> 32)) << 32) | (uint32_t)TH_UNI((int)(unsigned long long)(p))))
// compiler fences on single values: TH_OPAQUE_V(x) makes x (in a VGPR) opaque to the optimiser from here on, TH_KEEP_V(x) is a use of x
// at this point (loads that feed it are issued before it); neither emits an instruction
#define TH_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define TH_KEEP_V(x) asm volatile("" : : "v"(x))
#define TH_OPAQUE_S(x) asm volatile("" : "+s"(x))      // the same for a wave-uniform value in scalar registers; being volatile it also pins its basic block (no if-conversion)
#define TH_NOINLINE __attribute__((noinline))
// index into an LDS array that may run past its logical end by a few entries (the reads stay inside the workgroup's LDS and their
// values are discarded): nothing to do on the hardware
#define TH_LDS_SLACK_INDEX(i, cap) (i)

#define TH_BALLOT(b) __builtin_amdgcn_ballot_w64(b)      // straight from the comparison's lane mask (__ballot goes through a 0/1 register)
// 32-bit store to a wave-uniform base (scalar register pair) plus a 32-bit per-lane byte offset: the addressing mode the compiler does
// not pick when it can hoist `base + offset` into a loop-invariant 64-bit register pair per lane (two VGPRs that the POA row loop does
// not have to spare)
#ifndef TH_ASM_STORE_PRE
#define TH_ASM_STORE_PRE ""
#endif
__device__ __forceinline__ void th_store_u32_saddr(const void *base, uint32_t off, uint32_t v)
{
    asm volatile(TH_ASM_STORE_PRE "global_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}
#define TH_STORE_U32_SADDR(base, off, v) th_store_u32_saddr((base), (uint32_t)(off), (v))
// the same for the low 16 bits of v
__device__ __forceinline__ void th_store_u16_saddr(const void *base, uint32_t off, uint32_t v)
{
    asm volatile(TH_ASM_STORE_PRE "global_store_short %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}
#define TH_STORE_U16_SADDR(base, off, v) th_store_u16_saddr((base), (uint32_t)(off), (v))
// An LDS read the compiler does not wait for: the value may be used only after TH_LDS_ASYNC_WAIT(value) -- which must come before
// ANY use or copy of it, loop-carried copies included -- so that a whole loop trip can pass between request and use.
__device__ __forceinline__ unsigned long long th_lds_read64_async(const void *p)
{
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
    return v;
}
#define TH_LDS_READ64_ASYNC(p) th_lds_read64_async(p)
#define TH_LDS_ASYNC_WAIT(v) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory")
// ... and the form a loop needs that requests the NEXT value while it works on this one: wait, then copy INSIDE the asm block into registers of their own.  (As a wait
// with a tied operand followed by a C++ copy, the allocator may place the copy in front of the wait -- it is free to, the asm being "some operation on a register" -- and
// did once the wait stood at the top of the trip loop: stale records, found by the 768-entry ring's GPU test; tools/isa_check_async_lds.py follows the control flow now.)
#define TH_LDS_ASYNC_TAKE(dst, v) asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mov_b64 %0, %1" : "=&v"(dst) : "v"(v) : "memory")
#define TH_LANE_OF_MASK(m, lane) __builtin_amdgcn_inverse_ballot_w64(m)      // this lane's bit of a wave-uniform mask: the mask IS the condition register
#define TH_CLZ_NZ(x) __builtin_clz((unsigned)(x))          // x != 0 where the result is used (garbage lanes are discarded by the caller)

typedef short th_s2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t th_pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(th_s2_t, a), __builtin_bit_cast(th_s2_t, b))); }
__device__ __forceinline__ uint32_t th_pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(th_s2_t, a), __builtin_bit_cast(th_s2_t, b))); }
__device__ __forceinline__ uint32_t th_pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(th_s2_t, a), __builtin_bit_cast(th_s2_t, b))); }
__device__ __forceinline__ uint32_t th_pk_negmask(uint32_t a) { const th_s2_t v = __builtin_bit_cast(th_s2_t, a); const th_s2_t r = v >> (th_s2_t)(15); return __builtin_bit_cast(uint32_t, r); }
// (as ONE instruction with the shift as an inline operand: the compiler turns the builtin into a v_perm_b32 whose selector is a 32-bit literal, and VOP3 takes no
// literal on gfx9 -- an s_mov_b32 per use.  The packed POA row ran nine such literal moves per trip, a fifth of its scalar instructions: round 5)
__device__ __forceinline__ uint32_t th_alignbit16(uint32_t hi, uint32_t lo) { uint32_t r; asm("v_alignbit_b32 %0, %1, %2, 16" : "=v"(r) : "v"(hi), "v"(lo)); return r; }
// both halves minus a constant in the range of the inline operands (0 .. 64), saturating: hipcc canonicalises sub_sat(x, K) to add_sat(x, -K), and -24 / -32
// are literals where 24 / 32 are inline
#define TH_PK_SUB_IMM(x, K) ({ uint32_t r_; asm("v_pk_sub_i16 %0, %1, %2 op_sel_hi:[1,0] clamp" : "=v"(r_) : "v"(x), "n"(K)); r_; })
// 0x8000 ("minus infinity") in every 16-bit half the mask m (0xffff / 0 per half) leaves out: ~m & 0x80008000 as two VOP2 instructions of the fast class, the
// constant a literal of the second -- the select (x & m) | that is then ONE v_and_or_b32 on three vector registers, where v_bfi_b32 against the constant
// needed it in a scalar register (moved there per use)
#ifndef TH_R5_NEGM_ASM
#define TH_R5_NEGM_ASM 1
#endif
__device__ __forceinline__ uint32_t th_pk_neg_outside(uint32_t m)
{
#if TH_R5_NEGM_ASM
    uint32_t r; asm("v_not_b32 %0, %1\n\tv_and_b32 %0, 0x80008000, %0" : "=&v"(r) : "v"(m)); return r;
#else
    return ~m & 0x80008000u;
#endif
}
// the byte-permute selector of the packed row's score look-up from a byte holding two 4-bit query codes: ((x << 12 | x) & 0x000f000f) | 0x0c000c00 with the two
// constants as VOP2 literals (fused into v_bitop3 / v_and_or they need scalar registers)
__device__ __forceinline__ uint32_t th_pk_score_sel(uint32_t qx)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, 12, %1\n\tv_and_b32 %0, 0x000f000f, %0\n\tv_or_b32 %0, 0x0c000c00, %0" : "=&v"(r) : "v"(qx));
    return r;
}
// the same as th_pk_negmask, but opaque to the optimiser: hipcc turns ashr(sub_sat(a, b), 15) feeding a bit-select into
// two 16-bit compares, two v_cndmask and a v_perm; written as one instruction it stays v_pk_sub / v_pk_ashrrev / v_bfi
__device__ __forceinline__ uint32_t th_pk_negmask_op(uint32_t a) { uint32_t r; asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
// sign bits of the two halves as 0 / 1 in bits 0 and 16: one v_pk_lshrrev_b16 (kept opaque for the same reason)
__device__ __forceinline__ uint32_t th_pk_sign(uint32_t a) { uint32_t r; asm("v_pk_lshrrev_b16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ uint32_t th_ubfe(uint32_t v, uint32_t off, uint32_t w) { return __builtin_amdgcn_ubfe(v, off, w); }
__device__ __forceinline__ uint32_t th_cvt_pk_i16(int lo, int hi) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(lo, hi)); }
__device__ __forceinline__ uint32_t th_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
// v_perm_b32 selectors 8..11: a byte of 0xff / 0x00 from the sign of S1[15], S1[31], S0[15], S0[31].  TH_PERM_SIGNS: the signs of the two
// 16-bit halves of s0 and s1 as bytes (s0.hi, s1.hi, s0.lo, s1.lo)
// set bits of a ballot below this lane (v_mbcnt_lo / _hi: no per-lane 64-bit mask to keep alive)
__device__ __forceinline__ int th_mbcnt(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
#define TH_PERM_SIGNS 0x0b090a08u
// ... and ordered by source: (s1.hi, s1.lo, s0.hi, s0.lo) -- the two halves of one difference side by side (the 8-bit direction words)
#define TH_PERM_SIGNS8 0x09080b0au
__device__ __forceinline__ uint32_t th_and_or(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }
__device__ __forceinline__ uint32_t th_wave_incl_pkmax(uint32_t v)
{
    const int id = (int)TH_PK_NEG;
    uint32_t x = v;
    x = th_pk_max(x, (uint32_t)TH_DPP(id, (int)v, 0x111, 0xf, 0xf));     // row_shr:1
    x = th_pk_max(x, (uint32_t)TH_DPP(id, (int)v, 0x112, 0xf, 0xf));     // row_shr:2
    x = th_pk_max(x, (uint32_t)TH_DPP(id, (int)v, 0x113, 0xf, 0xf));     // row_shr:3
    x = th_pk_max(x, (uint32_t)TH_DPP(id, (int)x, 0x114, 0xf, 0xe));     // row_shr:4, banks 1-3
    x = th_pk_max(x, (uint32_t)TH_DPP(id, (int)x, 0x118, 0xf, 0xc));     // row_shr:8, banks 2-3
    x = th_pk_max(x, (uint32_t)TH_DPP(id, (int)x, 0x142, 0xa, 0xf));     // row_bcast:15 into rows 1,3
    x = th_pk_max(x, (uint32_t)TH_DPP(id, (int)x, 0x143, 0xc, 0xf));     // row_bcast:31 into rows 2,3
    return x;
}

#endif
