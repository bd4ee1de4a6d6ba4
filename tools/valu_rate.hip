// tools/valu_rate.hip -- issue rate of the VALU instructions the DP kernels are made of (gfx950), development helper.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Every wave runs ITER x 16 independent instructions of one kind (16 accumulators, no dependent pairs closer than 16
// instructions apart); 1 / 2 / 8 waves per SIMD on every CU.  Prints cycles per wave-instruction per SIMD at the clock
// the chip held (s_memtime ticks of the kernel's first wave against the event time give the clock; 2.4 GHz is only the
// fallback).  Round 5: the float forms (DP values are integers below 2^24: exact in f32), the remaining integer forms,
// v_cndmask with an SGPR-pair mask, lane ops and the scalar side (s_and_b64 etc. for the ballot algebra).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define ITER 2048
#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(64) NAME(unsigned *out, unsigned seed, unsigned long long msk)       \
    {                                                                                                       \
        unsigned a[16];                                                                                     \
        for (int i = 0; i < 16; ++i) a[i] = __float_as_uint((float)(seed + threadIdx.x * 17 + i));          \
        unsigned b = __float_as_uint((float)(seed * 3 + threadIdx.x));                                      \
        unsigned long long m = msk;                                                                         \
        asm volatile("s_mov_b64 vcc, %0" ::"s"(m) : "vcc");                                                 \
        for (int it = 0; it < ITER; ++it) {                                                                 \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(8) ASM(9) ASM(10) ASM(11) ASM(12) ASM(13) ASM(14) ASM(15) \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),       \
                           "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])  \
                         : "v"(b), "s"(m)                                                                   \
                         : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");           \
        }                                                                                                   \
        unsigned s = 0;                                                                                     \
        for (int i = 0; i < 16; ++i) s ^= a[i];                                                             \
        out[blockIdx.x * 64 + threadIdx.x] = s;                                                             \
    }
// %i = accumulator i, %16 = a vector operand, %17 = an SGPR pair
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %16\n"
#define A_SUB(i) "v_sub_u32 %" #i ", %" #i ", %16\n"
#define A_SUBREV(i) "v_subrev_u32 %" #i ", %" #i ", %16\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %16\n"
#define A_OR(i) "v_or_b32 %" #i ", %" #i ", %16\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %16\n"
#define A_NOT(i) "v_not_b32 %" #i ", %" #i "\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %16\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n"
#define A_LSHLV(i) "v_lshlrev_b32 %" #i ", %16, %" #i "\n"
#define A_MAX(i) "v_max_i32 %" #i ", %" #i ", %16\n"
#define A_MIN(i) "v_min_i32 %" #i ", %" #i ", %16\n"
#define A_MAXU(i) "v_max_u32 %" #i ", %" #i ", %16\n"
#define A_MINU(i) "v_min_u32 %" #i ", %" #i ", %16\n"
#define A_MAX3(i) "v_max3_i32 %" #i ", %" #i ", %16, %16\n"
#define A_MIN3(i) "v_min3_i32 %" #i ", %" #i ", %16, %16\n"
#define A_MED3(i) "v_med3_i32 %" #i ", %" #i ", %16, %16\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %16, %16\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %16\n"
#define A_ADDLSHL(i) "v_add_lshl_u32 %" #i ", %" #i ", %16, 2\n"
#define A_XAD(i) "v_xad_u32 %" #i ", %" #i ", %16, %16\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %16, %16\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %16\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %16, %16\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
#define A_FFBH(i) "v_ffbh_u32 %" #i ", %" #i "\n"
#define A_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %16\n"
#define A_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %16\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %16, %" #i ", %16\n"
#define A_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 16\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %16, %16\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 16, %16\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %16, %16\n"
#define A_SAD(i) "v_sad_u32 %" #i ", %" #i ", %16, %16\n"
#define A_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %16\n"
#define A_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %16, vcc\n"
#define A_ADDK(i) "v_add_u32 %" #i ", 7, %" #i "\n"
#define A_ADDS(i) "v_add_u32 %" #i ", s17, %" #i "\n"
#define A_MAXK(i) "v_max_i32 %" #i ", 7, %" #i "\n"
#define A_ADDU16(i) "v_add_u16 %" #i ", %" #i ", %16\n"
#define A_MAXI16(i) "v_max_i16 %" #i ", %" #i ", %16\n"
#define A_PKADD(i) "v_pk_add_i16 %" #i ", %" #i ", %16 clamp\n"
#define A_PKADDNC(i) "v_pk_add_i16 %" #i ", %" #i ", %16\n"
#define A_PKADDU(i) "v_pk_add_u16 %" #i ", %" #i ", %16\n"
#define A_PKSUB(i) "v_pk_sub_i16 %" #i ", %" #i ", %16 clamp\n"
#define A_PKMAX(i) "v_pk_max_i16 %" #i ", %" #i ", %16\n"
#define A_PKMIN(i) "v_pk_min_i16 %" #i ", %" #i ", %16\n"
#define A_PKMAXU(i) "v_pk_max_u16 %" #i ", %" #i ", %16\n"
#define A_PKASHR(i) "v_pk_ashrrev_i16 %" #i ", 15, %" #i " op_sel_hi:[0,1]\n"
#define A_PKLSHL(i) "v_pk_lshlrev_b16 %" #i ", 3, %" #i " op_sel_hi:[0,1]\n"
#define A_PKMAD(i) "v_pk_mad_i16 %" #i ", %" #i ", %16, %16\n"
#define A_PKMULLO(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %16\n"
#define A_DPP(i) "v_mov_b32_dpp %" #i ", %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_DPPROW(i) "v_mov_b32_dpp %" #i ", %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_DPPROWBC(i) "v_mov_b32_dpp %" #i ", %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define A_DPPBCAST(i) "v_mov_b32_dpp %" #i ", %16 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define A_DPPQUAD(i) "v_mov_b32_dpp %" #i ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_MAXDPP(i) "v_max_i32_dpp %" #i ", %16, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_ADDDPP(i) "v_add_u32_dpp %" #i ", %16, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_XORDPP(i) "v_xor_b32_dpp %" #i ", %16, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
#define A_SDWA(i) "v_max_i32_sdwa %" #i ", sext(%" #i "), sext(%16) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define A_CNDMASKS(i) "v_cndmask_b32 %" #i ", %" #i ", %16, %17\n"
#define A_CMP(i) "v_cmp_gt_i32 vcc, %" #i ", %16\n"
#define A_CMPS(i) "v_cmp_gt_i32 s[40:41], %" #i ", %16\n"
#define A_CMPU(i) "v_cmp_gt_u32 vcc, %" #i ", %16\n"
#define A_CMPEQ(i) "v_cmp_eq_u32 vcc, %" #i ", %16\n"
#define A_CMPPAIR(i) "v_cmp_gt_i32 vcc, %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define A_READLANE(i) "v_readlane_b32 s40, %" #i ", 5\n"
#define A_READFIRST(i) "v_readfirstlane_b32 s40, %" #i "\n"
#define A_WRITELANE(i) "v_writelane_b32 %" #i ", s16, 5\n"
// float forms (all DP values are integers below 2^24)
#define A_ADDF(i) "v_add_f32 %" #i ", %" #i ", %16\n"
#define A_SUBF(i) "v_sub_f32 %" #i ", %" #i ", %16\n"
#define A_MULF(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
#define A_FMAF(i) "v_fma_f32 %" #i ", %" #i ", %16, %16\n"
#define A_FMACF(i) "v_fmac_f32 %" #i ", %16, %16\n"
#define A_MAXF(i) "v_max_f32 %" #i ", %" #i ", %16\n"
#define A_MINF(i) "v_min_f32 %" #i ", %" #i ", %16\n"
#define A_MAX3F(i) "v_max3_f32 %" #i ", %" #i ", %16, %16\n"
#define A_MIN3F(i) "v_min3_f32 %" #i ", %" #i ", %16, %16\n"
#define A_MED3F(i) "v_med3_f32 %" #i ", %" #i ", %16, %16\n"
#define A_CMPF(i) "v_cmp_gt_f32 vcc, %" #i ", %16\n"
#define A_MAXFDPP(i) "v_max_f32_dpp %" #i ", %16, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_ADDFDPP(i) "v_add_f32_dpp %" #i ", %16, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_CVTFI(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define A_CVTIF(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define A_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_PKADDF16(i) "v_pk_add_f16 %" #i ", %" #i ", %16\n"
#define A_PKMAXF16(i) "v_pk_max_f16 %" #i ", %" #i ", %16\n"
#define A_PKFMAF16(i) "v_pk_fma_f16 %" #i ", %" #i ", %16, %16\n"
#define A_MAXF16(i) "v_max_f16 %" #i ", %" #i ", %16\n"
// scalar side: what the ballot algebra of the chain DP and the POA's row dispatch are made of
#define A_SAND64(i) "s_and_b64 s[40:41], s[40:41], %17\n"
#define A_SANDN64(i) "s_andn2_b64 s[42:43], s[42:43], %17\n"
#define A_SADD(i) "s_add_u32 s44, s44, s45\n"
#define A_SFF1(i) "s_ff1_i32_b64 s44, %17\n"
#define A_SBCNT(i) "s_bcnt1_i32_b64 s44, %17\n"
#define A_SLSHL64(i) "s_lshl_b64 s[40:41], s[40:41], 1\n"
#define A_SCMP(i) "s_cmp_lt_u32 s44, s45\n"
#define A_SCSEL(i) "s_cselect_b32 s44, s44, s45\n"
#define A_SMOV(i) "s_mov_b32 s44, s45\n"
#define A_SNOP(i) "s_nop 0\n"
// mixes: does a scalar instruction cost a vector slot?  (alternating streams)
#define A_MIXVS(i) "v_max_i32 %" #i ", %" #i ", %16\ns_and_b64 s[40:41], s[40:41], %17\n"
#define A_MIXAS(i) "v_add_u32 %" #i ", %" #i ", %16\ns_and_b64 s[40:41], s[40:41], %17\n"
#define A_MIXAM(i) "v_add_u32 %" #i ", %" #i ", %16\nv_max_i32 %" #i ", %" #i ", %16\n"
#define A_MIXPKADDMAX(i) "v_pk_add_i16 %" #i ", %" #i ", %16 clamp\nv_pk_max_i16 %" #i ", %" #i ", %16\n"

#define LIST(X)                                                                                                                          \
    X(k_add, A_ADD, "v_add_u32", 1) X(k_sub, A_SUB, "v_sub_u32", 1) X(k_subrev, A_SUBREV, "v_subrev_u32", 1) X(k_and, A_AND, "v_and_b32", 1)  \
    X(k_or, A_OR, "v_or_b32", 1) X(k_xor, A_XOR, "v_xor_b32", 1) X(k_not, A_NOT, "v_not_b32", 1) X(k_mov, A_MOV, "v_mov_b32", 1)              \
    X(k_lshl, A_LSHL, "v_lshlrev_b32 imm", 1) X(k_lshr, A_LSHR, "v_lshrrev_b32 imm", 1) X(k_ashr, A_ASHR, "v_ashrrev_i32 imm", 1)             \
    X(k_lshlv, A_LSHLV, "v_lshlrev_b32 vgpr", 1) X(k_addk, A_ADDK, "v_add_u32 inline const", 1) X(k_adds, A_ADDS, "v_add_u32 sgpr", 1)        \
    X(k_max, A_MAX, "v_max_i32", 1) X(k_maxk, A_MAXK, "v_max_i32 inline const", 1) X(k_min, A_MIN, "v_min_i32", 1) X(k_maxu, A_MAXU, "v_max_u32", 1) \
    X(k_minu, A_MINU, "v_min_u32", 1) X(k_max3, A_MAX3, "v_max3_i32", 1) X(k_min3, A_MIN3, "v_min3_i32", 1) X(k_med3, A_MED3, "v_med3_i32", 1) \
    X(k_add3, A_ADD3, "v_add3_u32", 1) X(k_lshladd, A_LSHLADD, "v_lshl_add_u32", 1) X(k_addlshl, A_ADDLSHL, "v_add_lshl_u32", 1)              \
    X(k_xad, A_XAD, "v_xad_u32", 1) X(k_or3, A_OR3, "v_or3_b32", 1) X(k_mul24, A_MUL24, "v_mul_u32_u24", 1) X(k_mad24, A_MAD24, "v_mad_u32_u24", 1) \
    X(k_mullo, A_MULLO, "v_mul_lo_u32", 1) X(k_ffbh, A_FFBH, "v_ffbh_u32", 1) X(k_bcnt, A_BCNT, "v_bcnt_u32_b32", 1)                          \
    X(k_mbcnt, A_MBCNT, "v_mbcnt_lo_u32_b32", 1) X(k_bfe, A_BFE, "v_bfe_u32", 1) X(k_bfi, A_BFI, "v_bfi_b32", 1)                              \
    X(k_align, A_ALIGN, "v_alignbit_b32", 1) X(k_perm, A_PERM, "v_perm_b32", 1) X(k_lshlor, A_LSHLOR, "v_lshl_or_b32", 1)                     \
    X(k_andor, A_ANDOR, "v_and_or_b32", 1) X(k_sad, A_SAD, "v_sad_u32", 1) X(k_addco, A_ADDCO, "v_add_co_u32", 1) X(k_addc, A_ADDC, "v_addc_co_u32", 1) \
    X(k_addu16, A_ADDU16, "v_add_u16", 1) X(k_maxi16, A_MAXI16, "v_max_i16", 1)                                                                \
    X(k_pkadd, A_PKADD, "v_pk_add_i16 clamp", 1) X(k_pkaddnc, A_PKADDNC, "v_pk_add_i16", 1) X(k_pkaddu, A_PKADDU, "v_pk_add_u16", 1)          \
    X(k_pksub, A_PKSUB, "v_pk_sub_i16 clamp", 1) X(k_pkmax, A_PKMAX, "v_pk_max_i16", 1) X(k_pkmin, A_PKMIN, "v_pk_min_i16", 1)                \
    X(k_pkmaxu, A_PKMAXU, "v_pk_max_u16", 1) X(k_pkashr, A_PKASHR, "v_pk_ashrrev_i16", 1) X(k_pklshl, A_PKLSHL, "v_pk_lshlrev_b16", 1)        \
    X(k_pkmad, A_PKMAD, "v_pk_mad_i16", 1) X(k_pkmullo, A_PKMULLO, "v_pk_mul_lo_u16", 1)                                                       \
    X(k_dpp, A_DPP, "v_mov_b32_dpp wave_shr", 1) X(k_dpprow, A_DPPROW, "v_mov_b32_dpp row_shr", 1) X(k_dpprowbc, A_DPPROWBC, "v_mov_b32_dpp row_shr bound_ctrl", 1) \
    X(k_dppbcast, A_DPPBCAST, "v_mov_b32_dpp row_bcast15", 1) X(k_dppquad, A_DPPQUAD, "v_mov_b32_dpp quad_perm", 1)                            \
    X(k_maxdpp, A_MAXDPP, "v_max_i32_dpp row_shr", 1) X(k_adddpp, A_ADDDPP, "v_add_u32_dpp row_shr", 1) X(k_xordpp, A_XORDPP, "v_xor_b32_dpp row_shr", 1) \
    X(k_addsdwa, A_ADDSDWA, "v_add_u32_sdwa", 1) X(k_sdwa, A_SDWA, "v_max_i32_sdwa", 1)                                                        \
    X(k_cndmask, A_CNDMASK, "v_cndmask_b32 vcc", 1) X(k_cndmasks, A_CNDMASKS, "v_cndmask_b32 sgpr pair", 1)                                   \
    X(k_cmp, A_CMP, "v_cmp_gt_i32 vcc", 1) X(k_cmps, A_CMPS, "v_cmp_gt_i32 sgpr pair", 1) X(k_cmpu, A_CMPU, "v_cmp_gt_u32 vcc", 1)            \
    X(k_cmpeq, A_CMPEQ, "v_cmp_eq_u32 vcc", 1) X(k_cmppair, A_CMPPAIR, "v_cmp_gt_i32 + v_cndmask (pair)", 2)                                  \
    X(k_readlane, A_READLANE, "v_readlane_b32", 1) X(k_readfirst, A_READFIRST, "v_readfirstlane_b32", 1) X(k_writelane, A_WRITELANE, "v_writelane_b32", 1) \
    X(k_addf, A_ADDF, "v_add_f32", 1) X(k_subf, A_SUBF, "v_sub_f32", 1) X(k_mulf, A_MULF, "v_mul_f32", 1) X(k_fmaf, A_FMAF, "v_fma_f32", 1)   \
    X(k_fmacf, A_FMACF, "v_fmac_f32", 1) X(k_maxf, A_MAXF, "v_max_f32", 1) X(k_minf, A_MINF, "v_min_f32", 1) X(k_max3f, A_MAX3F, "v_max3_f32", 1) \
    X(k_min3f, A_MIN3F, "v_min3_f32", 1) X(k_med3f, A_MED3F, "v_med3_f32", 1) X(k_cmpf, A_CMPF, "v_cmp_gt_f32 vcc", 1)                        \
    X(k_maxfdpp, A_MAXFDPP, "v_max_f32_dpp row_shr", 1) X(k_addfdpp, A_ADDFDPP, "v_add_f32_dpp row_shr", 1)                                   \
    X(k_cvtfi, A_CVTFI, "v_cvt_f32_i32", 1) X(k_cvtif, A_CVTIF, "v_cvt_i32_f32", 1) X(k_cvtfu, A_CVTFU, "v_cvt_f32_u32", 1)                   \
    X(k_pkaddf16, A_PKADDF16, "v_pk_add_f16", 1) X(k_pkmaxf16, A_PKMAXF16, "v_pk_max_f16", 1) X(k_pkfmaf16, A_PKFMAF16, "v_pk_fma_f16", 1)    \
    X(k_maxf16, A_MAXF16, "v_max_f16", 1)                                                                                                      \
    X(k_sand64, A_SAND64, "s_and_b64", 1) X(k_sandn64, A_SANDN64, "s_andn2_b64", 1) X(k_sadd, A_SADD, "s_add_u32", 1) X(k_sff1, A_SFF1, "s_ff1_i32_b64", 1) \
    X(k_sbcnt, A_SBCNT, "s_bcnt1_i32_b64", 1) X(k_slshl64, A_SLSHL64, "s_lshl_b64", 1) X(k_scmp, A_SCMP, "s_cmp_lt_u32", 1)                   \
    X(k_scsel, A_SCSEL, "s_cselect_b32", 1) X(k_smov, A_SMOV, "s_mov_b32", 1) X(k_snop, A_SNOP, "s_nop 0", 1)                                  \
    X(k_mixvs, A_MIXVS, "v_max_i32 + s_and_b64 (pair)", 2) X(k_mixas, A_MIXAS, "v_add_u32 + s_and_b64 (pair)", 2)                              \
    X(k_mixam, A_MIXAM, "v_add_u32 + v_max_i32 (pair)", 2) X(k_mixpk, A_MIXPKADDMAX, "v_pk_add_i16 + v_pk_max_i16 (pair)", 2)

#define DEF(K, A, N, C) KERNEL(K, A)
LIST(DEF)

// shader clock: every wave's lane 0 reports its s_memtime span of a fixed-length v_add chain?  Simpler: a spin kernel that
// counts s_memtime ticks over a known wall time.
__global__ void k_clock(unsigned long long *out, int spins)
{
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long w0 = wall_clock64();
    unsigned x = threadIdx.x;
    for (int i = 0; i < spins; ++i) asm volatile("v_add_u32 %0, %0, %0\n" : "+v"(x));
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = x; }
}

static double g_ghz = 2.4;
template <typename K> static void run(const char *name, K kern, unsigned *d, int waves_per_simd, int per_slot)
{
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 1u, 0x5555aaaa3333ccccull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 2u, 0x5555aaaa3333ccccull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * ITER * 16 * per_slot;
    printf("%-36s %d waves/SIMD  %.3f ms  %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e-3 * g_ghz * 1e9 / instr_per_simd);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}
int main(int argc, char **argv)
{
    unsigned *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    unsigned long long *c; hipMalloc(&c, 64); unsigned long long h[3] = {0, 0, 0};
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, c, 2000000); hipDeviceSynchronize();
    hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
    // wall_clock64 ticks at 100 MHz on gfx9
    if (h[1]) { double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9; printf("# shader clock from s_memtime / wall_clock64 (100 MHz): %.3f GHz (one idle wave)\n", ghz); }
    printf("# cycles assume %.2f GHz\n", g_ghz);
    const char *only = argc > 1 && argv[1][0] ? argv[1] : nullptr;      // substring filter on the instruction name
    const char *skip = argc > 2 && argv[2][0] ? argv[2] : nullptr;      // ... and an exclusion
    for (int w : {1, 2, 8}) {
#define RUN(K, A, N, C) if ((!only || strstr(N, only)) && !(skip && strstr(N, skip))) run(N, K, d, w, C);
        LIST(RUN)
    }
    return 0;
}
