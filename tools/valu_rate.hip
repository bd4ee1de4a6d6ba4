// tools/valu_rate.hip -- issue rate of the VALU instructions the DP kernels are made of (gfx950), development helper.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Every wave runs ITER x 16 independent instructions of one kind (16 accumulators, no dependent pairs closer than 16
// instructions apart); 8 waves per SIMD on every CU.  Prints cycles per wave-instruction per SIMD (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 2048
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(64) NAME(unsigned *out, unsigned seed)                               \
    {                                                                                                       \
        unsigned a[16];                                                                                     \
        for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 17 + i;                                    \
        unsigned b = seed * 3 + threadIdx.x;                                                                \
        for (int it = 0; it < ITER; ++it) {                                                                 \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(8) ASM(9) ASM(10) ASM(11) ASM(12) ASM(13) ASM(14) ASM(15) \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),       \
                           "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])  \
                         : "v"(b));                                                                         \
        }                                                                                                   \
        unsigned s = 0;                                                                                     \
        for (int i = 0; i < 16; ++i) s ^= a[i];                                                             \
        out[blockIdx.x * 64 + threadIdx.x] = s;                                                             \
    }
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %16\n"
#define A_MAX(i) "v_max_i32 %" #i ", %" #i ", %16\n"
#define A_PKADD(i) "v_pk_add_i16 %" #i ", %" #i ", %16 clamp\n"
#define A_PKSUB(i) "v_pk_sub_i16 %" #i ", %" #i ", %16 clamp\n"
#define A_PKMAX(i) "v_pk_max_i16 %" #i ", %" #i ", %16\n"
#define A_PKASHR(i) "v_pk_ashrrev_i16 %" #i ", 15, %" #i " op_sel_hi:[0,1]\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %16, %" #i ", %16\n"
#define A_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 16\n"
#define A_DPP(i) "v_mov_b32_dpp %" #i ", %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_DPPROW(i) "v_mov_b32_dpp %" #i ", %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_MAXDPP(i) "v_max_i32_dpp %" #i ", %16, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_MAX3(i) "v_max3_i32 %" #i ", %" #i ", %16, %16\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define A_CMP(i) "v_cmp_gt_i32 vcc, %" #i ", %16\n"
#define A_SDWA(i) "v_max_i32_sdwa %" #i ", sext(%" #i "), sext(%16) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %16, %16\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 16, %16\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %16, %16\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %16\n"
KERNEL(k_add, A_ADD) KERNEL(k_max, A_MAX) KERNEL(k_pkadd, A_PKADD) KERNEL(k_pksub, A_PKSUB) KERNEL(k_pkmax, A_PKMAX) KERNEL(k_pkashr, A_PKASHR)
KERNEL(k_bfi, A_BFI) KERNEL(k_align, A_ALIGN) KERNEL(k_dpp, A_DPP) KERNEL(k_dpprow, A_DPPROW) KERNEL(k_maxdpp, A_MAXDPP) KERNEL(k_max3, A_MAX3)
KERNEL(k_cndmask, A_CNDMASK) KERNEL(k_cmp, A_CMP) KERNEL(k_sdwa, A_SDWA) KERNEL(k_perm, A_PERM) KERNEL(k_lshlor, A_LSHLOR) KERNEL(k_andor, A_ANDOR) KERNEL(k_xor, A_XOR)

template <typename K> static void run(const char *name, K kern, unsigned *d, int waves_per_simd)
{
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * ITER * 16;
    printf("%-22s %d waves/SIMD  %.3f ms  %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    for (int w : {1, 2, 8}) {
        run("v_add_u32", k_add, d, w); run("v_max_i32", k_max, d, w); run("v_max3_i32", k_max3, d, w); run("v_xor_b32", k_xor, d, w);
        run("v_pk_add_i16 clamp", k_pkadd, d, w); run("v_pk_sub_i16 clamp", k_pksub, d, w); run("v_pk_max_i16", k_pkmax, d, w); run("v_pk_ashrrev_i16", k_pkashr, d, w);
        run("v_bfi_b32", k_bfi, d, w); run("v_alignbit_b32", k_align, d, w); run("v_perm_b32", k_perm, d, w); run("v_lshl_or_b32", k_lshlor, d, w); run("v_and_or_b32", k_andor, d, w);
        run("v_mov_b32_dpp wave_shr", k_dpp, d, w); run("v_mov_b32_dpp row_shr", k_dpprow, d, w); run("v_max_i32_dpp row_shr", k_maxdpp, d, w);
        run("v_cndmask_b32", k_cndmask, d, w); run("v_cmp_gt_i32", k_cmp, d, w); run("v_max_i32_sdwa", k_sdwa, d, w);
    }
    return 0;
}
