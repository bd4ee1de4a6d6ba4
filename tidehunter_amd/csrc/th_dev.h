// th_dev.h -- device-side common definitions and wave-level helpers (gfx950: 64-lane waves).
#pragma once
#include "th_rt.h"

#define TH_WAVE 64
#define TH_NEG (-0x20000000)

// Parameters the kernels read (a device-friendly projection of mini_tandem_para, src/tidehunter.h:47-61)
struct ThDevPara {
    int k;                       // k-mer length (-k)
    int w;                       // -w: minimizer window (1 = every k-mer)
    int w_seq;                   // TH_MINIMIZER_SEQ=1: minimizer selection by the sequential replay only (the tests' cross-check)
    int hpc;                     // -H: homopolymer-compressed k-mers (src/tandem_hit.c:46)
    uint32_t min_p, max_p;       // period limits, cast as at src/tandem_hit.c:235
    double one_minus_div;        // 1 - max_div (src/partition.c:202,252: the one double product on the path)
    int min_copy;
    int match, mismatch, go1, ge1, go2, ge2;   // POA scoring (src/abpoa_cons.c:12-28)
};

// one accepted chain (chain_t, src/tandem_chain.h:30-34) as the extract kernel emits it
struct ThChainRec { int read, cell_off, len, est_start, est_period, first_end, last_start, pad; };

// one consensus job = one run of >= min_copy+1 boundaries between -1 separators (src/gen_cons.c:192-205)
struct ThTask {
    int read;        // read index in the batch
    int pos_off;     // first boundary of the run in the compact par_pos array
    int pos_n;       // boundaries in the run (units = pos_n - 1)
    int unit_off;    // first POA unit in the unit table
    int n_units;     // units that pass the filter at src/abpoa_cons.c:43 (n_seqs)
    int sum_len, max_len;
    int cons_off;    // consensus / coverage output offset
    int cons_cap;
    int aln_off;     // first entry of this task in the per-unit iden_n array
    int narrow;      // 1: the job starts on the int16 POA variant (th_poa_job_is_narrow), 0: on the 32-bit one
    int noband;      // host, slow path: the job is being rerun without the adaptive band
};
struct ThUnit { int start, len; };   // start = index of the unit's first base inside the read

// The pipeline's device counters: one block, zeroed by th_k_encode at the start of a batch.  Kernels get pointers to single fields.
struct ThCnt {
    int rec;                 // chain records in the arena (th_k_chain_extract*)
    int err;                 // error flags (TH_E_*)
    int q_poa_n, q_poa_w;    // work-queue heads: POA int16 variant / 32-bit variant
    int q_aln_s;             // ... alignment items, small staging area
    int par_top;             // top of the par_pos arena (th_k_partition)
    int q_xw, q_part;        // ... reads of the extraction waves, chains of the partition waves
    int min_fb;              // reads whose minimizer selection fell back to the sequential replay
    int q_aln_m, q_aln_b;    // ... alignment items, mid / big staging area
    int n_retry;             // jobs the POA gave back (status 1-4, 7): the host takes the slow path when this is not 0
    int n_failed;            // jobs that stay without a consensus (status 5, 6)
    int pad0[3];
    // job build (the first six in the column order of th_k_jobs_count: the scan writes them in one go)
    int par_total, n_task, n_unit, cons_total, aln_total, n_task_n;
    int max_unit, max_cons_cap;
    unsigned long long need_n, need_w;       // POA scratch ints per wave the largest job of either variant asks for
    // item build
    int n_item_s, n_item_m, n_item_b, cc_total, max_cl, pad1[3];
    int pad2[28];
};
static_assert(sizeof(ThCnt) == 256, "counter block");
#define TH_E_REC_OVERFLOW 1       // chain record arena too small (rerun the extraction with a larger one)
#define TH_E_PART_OVERFLOW 2      // a par_pos region or the boundary buffer of the partition was too small
#define TH_E_ALN_BND 4            // boundary buffer of the alignment kernel too small
#define TH_E_MINIMIZER 8          // minimizer selection repeated a word or overflowed its buffer
#define TH_E_ALN_VERIFY 64        // TH_ALN_MODE=2: a banded alignment differs from the full matrix
#define TH_E_MSA_UNITS 128        // a job with fewer than two units ("Not enough sequences to perform msa.", src/abpoa_cons.c:61)
#define TH_E_JOB_CAP 256          // job / unit / par_pos tables too small (rerun the build with larger ones)


// Point where data written by some lanes of a wave (typically lane 0 after a sequential section) must be
// visible to the other lanes.  On the hardware the wave runs in lock-step, so program order plus a
// workgroup-scope fence is enough; the CPU emulation (tests only) runs lanes as independent fibers and
// needs a real rendezvous here.
#ifdef TH_CPU_EMUL
#define TH_WAVE_FENCE() do { (void)hipemul::wave_exchange(0); } while (0)
#elif defined(TH_SAFE_FENCE)
#define TH_WAVE_FENCE() __threadfence_block()
#else
// Every team in these kernels is ONE wavefront, so the hand-offs are lane-to-lane inside a wave: the
// wave's vector-memory instructions reach its CU's L1 in issue order, which already orders a lane's
// store before another lane's later load.  A wavefront-scope fence keeps the compiler from reordering
// and, unlike __threadfence_block(), does not stall on s_waitcnt vmcnt(0) (the store round trip to L2,
// ~1-2 us) -- that wait per POA row was the dominant cost of the first version.
#define TH_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
#endif

// ---- wave helpers (every lane of the wave must call these; control flow around them is wave-uniform) ----
// On gfx950 the lane shifts, scans and reductions are DPP moves on the VALU (row_shr / row_bcast /
// wave_shr: a few cycles each) instead of ds_bpermute through the LDS crossbar (~100 cycles of
// latency per hop, which made every dependent scan step a stall).  The CPU emulation (tests only)
// implements the same semantics with its rendezvous shuffles.
#ifdef TH_CPU_EMUL
#define TH_READLANE(v, l) __shfl((v), (l))
#define TH_WRITELANE(old, v, l) (TH_LANE() == (l) ? (v) : (old))
#define th_writelane6(a, b, c, d, e, f, va, vb, vc, vd, ve, vf, l) do { if (TH_LANE() == (l)) { a = (va); b = (vb); c = (vc); d = (vd); e = (ve); f = (vf); } } while (0)
__device__ __forceinline__ int th_shr1(int v, int fill) { int o = __shfl_up(v, 1); return TH_LANE() == 0 ? fill : o; }
__device__ __forceinline__ int th_shl1(int v, int fill) { int o = __shfl_down(v, 1); return TH_LANE() == 63 ? fill : o; }
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
#else
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
#endif
#define TH_INT_MIN (-0x7fffffff - 1)
__device__ __forceinline__ int th_wave_excl_max(int v, int seed) { return max(th_shr1(th_wave_incl_max(v), seed), seed); }
__device__ __forceinline__ int th_wave_max(int v) { return TH_READLANE(th_wave_incl_max(v), 63); }
__device__ __forceinline__ int th_wave_min(int v) { return -TH_READLANE(th_wave_incl_max(-v), 63); }
__device__ __forceinline__ int th_wave_sum(int v) { return TH_READLANE(th_wave_incl_sum(v), 63); }
// lane l <- lane l+off for a wave-uniform offset (fill where the source lane does not exist)
__device__ __forceinline__ int th_lane_shift(int v, int off, int fill)
{
    if (off == 0) return v;
    if (off == -1) return th_shr1(v, fill);
    if (off == 1) return th_shl1(v, fill);
    if (off == 2) return th_shl1(th_shl1(v, fill), fill);
    const int src = TH_LANE() + off;
    const int o = __shfl(v, src & 63);
    return (src >= 0 && src < TH_WAVE) ? o : fill;
}
// Plain load of a word that was updated with device-scope atomics earlier in the same kernel: atomics execute
// in L2 and do not refresh the CU's vector L1, so the read-back must bypass L1 (global_load ... sc1).
#ifdef TH_CPU_EMUL
#define TH_LOAD_L2(p) (*(p))
#else
#define TH_LOAD_L2(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

// Explicit "all my vector-memory loads have landed".  gfx9 counts loads AND stores in vmcnt, and the compiler places its
// wait at the first use of a loaded register; when that use sits in a loop whose body also issues fire-and-forget
// stores, the wait it inserts there drains the stores of the previous trip on every trip.  Waiting right after the
// (rare) loads instead leaves the steady-state trips without any vmcnt wait.
#ifdef TH_CPU_EMUL
#define TH_VM_WAIT0() ((void)0)
#define TH_WAVES_PER_SIMD(n)
#else
#define TH_VM_WAIT0() __builtin_amdgcn_s_waitcnt(0x0F70)      // vmcnt(0), expcnt and lgkmcnt untouched
#define TH_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))   // caps the VGPR budget so that n waves fit a SIMD
#endif
// next index of a shared work queue, wave-uniform.  Every lane takes part in the atomic (lane 0 adds 1, the
// others 0), so there is no divergent region around it: with the add under `if (lane == 0)` hipcc 7.2
// restructured the surrounding persistent loop into one that never fetched a second item.
__device__ __forceinline__ int th_queue_pop(int *queue)
{
    const int v = atomicAdd(queue, TH_LANE() == 0 ? 1 : 0);
    return TH_READLANE(v, 0);
}
// A value that is the same in every lane but that the compiler cannot prove uniform (it came from a per-lane load or an LDS
// read): v_readfirstlane moves it to an SGPR, so that what depends on it is scalar code and scalar branches instead of
// vector code under execution masks.
#ifdef TH_CPU_EMUL
#define TH_UNI(x) (x)
#else
#define TH_UNI(x) __builtin_amdgcn_readfirstlane(x)
#endif
__device__ __forceinline__ int th_ctz64(unsigned long long m) { return m ? __ffsll((long long)m) - 1 : 64; }
__device__ __forceinline__ int th_msb64(unsigned long long m) { return 63 - __clzll((long long)m); }   // m != 0
#ifdef TH_CPU_EMUL
#define TH_BALLOT(b) __ballot(b)
#define TH_STORE_U32_SADDR(base, off, v) (*(uint32_t *)((char *)(base) + (uint32_t)(off)) = (v))
#define TH_LDS_READ64_ASYNC(p) (*(p))
#define TH_LDS_ASYNC_WAIT(v) ((void)0)
#define TH_LANE_OF_MASK(m, lane) (((m) >> (lane)) & 1ull)
#define TH_CLZ_NZ(x) ((unsigned)(x) ? __builtin_clz((unsigned)(x)) : 32)
static inline int __mul24(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
#else
#define TH_BALLOT(b) __builtin_amdgcn_ballot_w64(b)      // straight from the comparison's lane mask (__ballot goes through a 0/1 register)
// 32-bit store to a wave-uniform base (scalar register pair) plus a 32-bit per-lane byte offset: the addressing mode the compiler does
// not pick when it can hoist `base + offset` into a loop-invariant 64-bit register pair per lane (two VGPRs that the POA row loop does
// not have to spare)
__device__ __forceinline__ void th_store_u32_saddr(const void *base, uint32_t off, uint32_t v)
{
    asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}
#define TH_STORE_U32_SADDR(base, off, v) th_store_u32_saddr((base), (uint32_t)(off), (v))
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
#define TH_LANE_OF_MASK(m, lane) __builtin_amdgcn_inverse_ballot_w64(m)      // this lane's bit of a wave-uniform mask: the mask IS the condition register
#define TH_CLZ_NZ(x) __builtin_clz((unsigned)(x))          // x != 0 where the result is used (garbage lanes are discarded by the caller)
#endif
// the same for masks known to be non-zero: bare s_ff1 / s_flbit, no zero fix-ups
__device__ __forceinline__ int th_ctz64nz(unsigned long long m) { return __builtin_ctzll(m); }
__device__ __forceinline__ int th_msb64nz(unsigned long long m) { return 63 - __builtin_clzll(m); }

// ---- packed int16 pairs (two DP cells per lane): v_pk_add_i16 clamp / v_pk_sub_i16 clamp / v_pk_max_i16 ----
#define TH_PK_NEG 0x80008000u            // (-32768, -32768): "minus infinity" in both halves
#define TH_PK_LO(x) ((int)(short)((x) & 0xffffu))
#define TH_PK_HI(x) ((int)(short)((uint32_t)(x) >> 16))
#define TH_PK(lo, hi) (((uint32_t)(lo) & 0xffffu) | ((uint32_t)(hi) << 16))
#define TH_PK_DUP(v) TH_PK((v), (v))
#ifdef TH_CPU_EMUL
static inline int th_sat16_(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
static inline uint32_t th_pk_add(uint32_t a, uint32_t b) { return TH_PK(th_sat16_(TH_PK_LO(a) + TH_PK_LO(b)), th_sat16_(TH_PK_HI(a) + TH_PK_HI(b))); }
static inline uint32_t th_pk_sub(uint32_t a, uint32_t b) { return TH_PK(th_sat16_(TH_PK_LO(a) - TH_PK_LO(b)), th_sat16_(TH_PK_HI(a) - TH_PK_HI(b))); }
static inline uint32_t th_pk_max(uint32_t a, uint32_t b) { return TH_PK(TH_PK_LO(a) > TH_PK_LO(b) ? TH_PK_LO(a) : TH_PK_LO(b), TH_PK_HI(a) > TH_PK_HI(b) ? TH_PK_HI(a) : TH_PK_HI(b)); }
static inline uint32_t th_pk_negmask(uint32_t a) { return (TH_PK_LO(a) < 0 ? 0xffffu : 0u) | (TH_PK_HI(a) < 0 ? 0xffff0000u : 0u); }
static inline uint32_t th_alignbit16(uint32_t hi, uint32_t lo) { return (lo >> 16) | (hi << 16); }   // (lo.hi, hi.lo)
static inline uint32_t th_pk_negmask_op(uint32_t a) { return th_pk_negmask(a); }
static inline uint32_t th_pk_sign(uint32_t a) { return (TH_PK_LO(a) < 0 ? 1u : 0u) | (TH_PK_HI(a) < 0 ? 0x10000u : 0u); }
static inline uint32_t th_ubfe(uint32_t v, uint32_t off, uint32_t w) { return (v >> (off & 31)) & ((1u << w) - 1); }
static inline uint32_t th_cvt_pk_i16(int lo, int hi) { return TH_PK(th_sat16_(lo), th_sat16_(hi)); }     // two ints to a saturated int16 pair
// v_perm_b32: result byte k is picked by selector byte k: 0..3 byte of s1, 4..7 byte of s0, 12 the constant 0 (the other codes are not used here)
static inline uint32_t th_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const unsigned long long src = ((unsigned long long)s0 << 32) | s1; uint32_t r = 0;
    for (int k = 0; k < 4; ++k) { const uint32_t c = (sel >> (8 * k)) & 0xff; r |= (c < 8 ? (uint32_t)((src >> (8 * c)) & 0xff) : c >= 13 ? 0xffu : 0u) << (8 * k); }
    return r;
}
__device__ __forceinline__ uint32_t th_wave_incl_pkmax(uint32_t v)
{
    const int lane = TH_LANE();
    for (int d = 1; d < TH_WAVE; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, d); if (lane >= d) v = th_pk_max(v, o); }
    return v;
}
#else
typedef short th_s2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t th_pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(th_s2_t, a), __builtin_bit_cast(th_s2_t, b))); }
__device__ __forceinline__ uint32_t th_pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(th_s2_t, a), __builtin_bit_cast(th_s2_t, b))); }
__device__ __forceinline__ uint32_t th_pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(th_s2_t, a), __builtin_bit_cast(th_s2_t, b))); }
__device__ __forceinline__ uint32_t th_pk_negmask(uint32_t a) { const th_s2_t v = __builtin_bit_cast(th_s2_t, a); const th_s2_t r = v >> (th_s2_t)(15); return __builtin_bit_cast(uint32_t, r); }
__device__ __forceinline__ uint32_t th_alignbit16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
// the same as th_pk_negmask, but opaque to the optimiser: hipcc turns ashr(sub_sat(a, b), 15) feeding a bit-select into
// two 16-bit compares, two v_cndmask and a v_perm; written as one instruction it stays v_pk_sub / v_pk_ashrrev / v_bfi
__device__ __forceinline__ uint32_t th_pk_negmask_op(uint32_t a) { uint32_t r; asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
// sign bits of the two halves as 0 / 1 in bits 0 and 16: one v_pk_lshrrev_b16 (kept opaque for the same reason)
__device__ __forceinline__ uint32_t th_pk_sign(uint32_t a) { uint32_t r; asm("v_pk_lshrrev_b16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ uint32_t th_ubfe(uint32_t v, uint32_t off, uint32_t w) { return __builtin_amdgcn_ubfe(v, off, w); }
__device__ __forceinline__ uint32_t th_cvt_pk_i16(int lo, int hi) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(lo, hi)); }
__device__ __forceinline__ uint32_t th_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
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
// ---- workgroup bitonic sort of N = 2^k words in LDS (or HBM), ascending or descending; ends with a barrier ----
// A compare-exchange stage costs a round trip through the buffer and a barrier, so stages are grouped: a thread loads 32
// words into registers, runs up to five consecutive stages there and stores them.  The grouped pass with unit u takes the words
// base + x u (x = 0..31; consecutive lanes take consecutive bases, so the strided LDS accesses do not collide) and covers
// the strides 16 u .. u that are <= the current one: u = 32 for the strides 512 .. 32, u = 1 for 16 .. 1.  A merge of size
// 2^k thus costs max(k - 10, 0) single passes + 2 grouped ones instead of k (33 passes instead of 105 at 16384 words).
template <typename W, bool DESC, int BLOCK>
__device__ __forceinline__ void th_bitonic_sort(W *buf, int N, int tid)
{
    auto reg_pass = [&](int u, int size, int stride) {
        for (int w = tid; w < (N >> 5); w += BLOCK) {
            W v[32];
            const int base = (w / u) * 32 * u + (w & (u - 1));
#pragma unroll
            for (int x = 0; x < 32; ++x) v[x] = buf[base + x * u];
#pragma unroll
            for (int st = 16; st > 0; st >>= 1) {
                if (st * u <= stride) {
#pragma unroll
                    for (int x = 0; x < 32; ++x) {
                        if ((x & st) == 0) {
                            const bool first = (((base + x * u) & size) == 0) != DESC;      // the smaller word goes first
                            const W a = v[x], c = v[x + st];
                            const W mn = a < c ? a : c, mx = a < c ? c : a;
                            v[x] = first ? mn : mx; v[x + st] = first ? mx : mn;
                        }
                    }
                }
            }
#pragma unroll
            for (int x = 0; x < 32; ++x) buf[base + x * u] = v[x];
        }
        __syncthreads();
    };
    const int top = N >= 1024 ? 512 : N >= 32 ? 16 : 0;      // largest stride the grouped passes take
    for (int size = 2; size <= N; size <<= 1) {
        int stride = size >> 1;
        for (; stride > top; stride >>= 1) {
            for (int t = tid; t < (N >> 1); t += BLOCK) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool first = ((lo & size) == 0) != DESC;
                const W a = buf[lo], c = buf[hi];
                if ((a > c) == first) { buf[lo] = c; buf[hi] = a; }
            }
            __syncthreads();
        }
        if (N >= 1024 && stride >= 32) { reg_pass(32, size, stride); stride = 16; }
        if (N >= 32 && stride >= 1) reg_pass(1, size, stride);
    }
}

// (a & m) | (b & ~m): v_bfi_b32
__device__ __forceinline__ uint32_t th_sel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
