// VALU issue-rate probe (third set) for gfx950: which instructions run at the 2-cycle rate of the fp32 FMA pipe and which at
// 4 cycles per wave64.  8 independent chains per lane, W waves per SIMD; prints shader cycles per wave64 instruction per SIMD
// (nominal 2.4 GHz clock).  Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate3.hip -o tools/ubench/valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b)
{
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = (float)threadIdx.x + j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define RUN(TEXT) asm volatile(TEXT : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc", "s20", "s21")
            if (KIND == 0) RUN("v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_fma_f32 %3, %3, %8, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_fma_f32 %6, %6, %8, %9\n" "v_fma_f32 %7, %7, %8, %9\n" );
            if (KIND == 1) RUN("v_mul_f32 %0, %0, %8\n" "v_mul_f32 %1, %1, %8\n" "v_mul_f32 %2, %2, %8\n" "v_mul_f32 %3, %3, %8\n" "v_mul_f32 %4, %4, %8\n" "v_mul_f32 %5, %5, %8\n" "v_mul_f32 %6, %6, %8\n" "v_mul_f32 %7, %7, %8\n" );
            if (KIND == 2) RUN("v_add_f32 %0, %0, %8\n" "v_add_f32 %1, %1, %8\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %8\n" "v_add_f32 %4, %4, %8\n" "v_add_f32 %5, %5, %8\n" "v_add_f32 %6, %6, %8\n" "v_add_f32 %7, %7, %8\n" );
            if (KIND == 3) RUN("v_sub_f32 %0, %0, %8\n" "v_sub_f32 %1, %1, %8\n" "v_sub_f32 %2, %2, %8\n" "v_sub_f32 %3, %3, %8\n" "v_sub_f32 %4, %4, %8\n" "v_sub_f32 %5, %5, %8\n" "v_sub_f32 %6, %6, %8\n" "v_sub_f32 %7, %7, %8\n" );
            if (KIND == 4) RUN("v_max_f32 %0, %0, %8\n" "v_max_f32 %1, %1, %8\n" "v_max_f32 %2, %2, %8\n" "v_max_f32 %3, %3, %8\n" "v_max_f32 %4, %4, %8\n" "v_max_f32 %5, %5, %8\n" "v_max_f32 %6, %6, %8\n" "v_max_f32 %7, %7, %8\n" );
            if (KIND == 5) RUN("v_min_f32 %0, %0, %8\n" "v_min_f32 %1, %1, %8\n" "v_min_f32 %2, %2, %8\n" "v_min_f32 %3, %3, %8\n" "v_min_f32 %4, %4, %8\n" "v_min_f32 %5, %5, %8\n" "v_min_f32 %6, %6, %8\n" "v_min_f32 %7, %7, %8\n" );
            if (KIND == 6) RUN("v_max3_f32 %0, %0, %8, %9\n" "v_max3_f32 %1, %1, %8, %9\n" "v_max3_f32 %2, %2, %8, %9\n" "v_max3_f32 %3, %3, %8, %9\n" "v_max3_f32 %4, %4, %8, %9\n" "v_max3_f32 %5, %5, %8, %9\n" "v_max3_f32 %6, %6, %8, %9\n" "v_max3_f32 %7, %7, %8, %9\n" );
            if (KIND == 7) RUN("v_cndmask_b32 %0, %0, %8, vcc\n" "v_cndmask_b32 %1, %1, %8, vcc\n" "v_cndmask_b32 %2, %2, %8, vcc\n" "v_cndmask_b32 %3, %3, %8, vcc\n" "v_cndmask_b32 %4, %4, %8, vcc\n" "v_cndmask_b32 %5, %5, %8, vcc\n" "v_cndmask_b32 %6, %6, %8, vcc\n" "v_cndmask_b32 %7, %7, %8, vcc\n" );
            if (KIND == 8) RUN("v_cmp_lt_f32 vcc, %0, %8\n" "v_cmp_lt_f32 vcc, %1, %8\n" "v_cmp_lt_f32 vcc, %2, %8\n" "v_cmp_lt_f32 vcc, %3, %8\n" "v_cmp_lt_f32 vcc, %4, %8\n" "v_cmp_lt_f32 vcc, %5, %8\n" "v_cmp_lt_f32 vcc, %6, %8\n" "v_cmp_lt_f32 vcc, %7, %8\n" );
            if (KIND == 9) RUN("v_cmp_lt_f32 s[20:21], %0, %8\n" "v_cmp_lt_f32 s[20:21], %1, %8\n" "v_cmp_lt_f32 s[20:21], %2, %8\n" "v_cmp_lt_f32 s[20:21], %3, %8\n" "v_cmp_lt_f32 s[20:21], %4, %8\n" "v_cmp_lt_f32 s[20:21], %5, %8\n" "v_cmp_lt_f32 s[20:21], %6, %8\n" "v_cmp_lt_f32 s[20:21], %7, %8\n" );
            if (KIND == 10) RUN("v_add_u32 %0, %0, %8\n" "v_add_u32 %1, %1, %8\n" "v_add_u32 %2, %2, %8\n" "v_add_u32 %3, %3, %8\n" "v_add_u32 %4, %4, %8\n" "v_add_u32 %5, %5, %8\n" "v_add_u32 %6, %6, %8\n" "v_add_u32 %7, %7, %8\n" );
            if (KIND == 11) RUN("v_sub_u32 %0, %0, %8\n" "v_sub_u32 %1, %1, %8\n" "v_sub_u32 %2, %2, %8\n" "v_sub_u32 %3, %3, %8\n" "v_sub_u32 %4, %4, %8\n" "v_sub_u32 %5, %5, %8\n" "v_sub_u32 %6, %6, %8\n" "v_sub_u32 %7, %7, %8\n" );
            if (KIND == 12) RUN("v_or_b32 %0, %0, %8\n" "v_or_b32 %1, %1, %8\n" "v_or_b32 %2, %2, %8\n" "v_or_b32 %3, %3, %8\n" "v_or_b32 %4, %4, %8\n" "v_or_b32 %5, %5, %8\n" "v_or_b32 %6, %6, %8\n" "v_or_b32 %7, %7, %8\n" );
            if (KIND == 13) RUN("v_xor_b32 %0, %0, %8\n" "v_xor_b32 %1, %1, %8\n" "v_xor_b32 %2, %2, %8\n" "v_xor_b32 %3, %3, %8\n" "v_xor_b32 %4, %4, %8\n" "v_xor_b32 %5, %5, %8\n" "v_xor_b32 %6, %6, %8\n" "v_xor_b32 %7, %7, %8\n" );
            if (KIND == 14) RUN("v_lshlrev_b32 %0, 3, %0\n" "v_lshlrev_b32 %1, 3, %1\n" "v_lshlrev_b32 %2, 3, %2\n" "v_lshlrev_b32 %3, 3, %3\n" "v_lshlrev_b32 %4, 3, %4\n" "v_lshlrev_b32 %5, 3, %5\n" "v_lshlrev_b32 %6, 3, %6\n" "v_lshlrev_b32 %7, 3, %7\n" );
            if (KIND == 15) RUN("v_lshrrev_b32 %0, 3, %0\n" "v_lshrrev_b32 %1, 3, %1\n" "v_lshrrev_b32 %2, 3, %2\n" "v_lshrrev_b32 %3, 3, %3\n" "v_lshrrev_b32 %4, 3, %4\n" "v_lshrrev_b32 %5, 3, %5\n" "v_lshrrev_b32 %6, 3, %6\n" "v_lshrrev_b32 %7, 3, %7\n" );
            if (KIND == 16) RUN("v_ashrrev_i32 %0, 3, %0\n" "v_ashrrev_i32 %1, 3, %1\n" "v_ashrrev_i32 %2, 3, %2\n" "v_ashrrev_i32 %3, 3, %3\n" "v_ashrrev_i32 %4, 3, %4\n" "v_ashrrev_i32 %5, 3, %5\n" "v_ashrrev_i32 %6, 3, %6\n" "v_ashrrev_i32 %7, 3, %7\n" );
            if (KIND == 17) RUN("v_lshl_or_b32 %0, %0, 6, %8\n" "v_lshl_or_b32 %1, %1, 6, %8\n" "v_lshl_or_b32 %2, %2, 6, %8\n" "v_lshl_or_b32 %3, %3, 6, %8\n" "v_lshl_or_b32 %4, %4, 6, %8\n" "v_lshl_or_b32 %5, %5, 6, %8\n" "v_lshl_or_b32 %6, %6, 6, %8\n" "v_lshl_or_b32 %7, %7, 6, %8\n" );
            if (KIND == 18) RUN("v_lshl_add_u32 %0, %0, 6, %8\n" "v_lshl_add_u32 %1, %1, 6, %8\n" "v_lshl_add_u32 %2, %2, 6, %8\n" "v_lshl_add_u32 %3, %3, 6, %8\n" "v_lshl_add_u32 %4, %4, 6, %8\n" "v_lshl_add_u32 %5, %5, 6, %8\n" "v_lshl_add_u32 %6, %6, 6, %8\n" "v_lshl_add_u32 %7, %7, 6, %8\n" );
            if (KIND == 19) RUN("v_and_or_b32 %0, %0, %8, %9\n" "v_and_or_b32 %1, %1, %8, %9\n" "v_and_or_b32 %2, %2, %8, %9\n" "v_and_or_b32 %3, %3, %8, %9\n" "v_and_or_b32 %4, %4, %8, %9\n" "v_and_or_b32 %5, %5, %8, %9\n" "v_and_or_b32 %6, %6, %8, %9\n" "v_and_or_b32 %7, %7, %8, %9\n" );
            if (KIND == 20) RUN("v_bfi_b32 %0, %8, %9, %0\n" "v_bfi_b32 %1, %8, %9, %1\n" "v_bfi_b32 %2, %8, %9, %2\n" "v_bfi_b32 %3, %8, %9, %3\n" "v_bfi_b32 %4, %8, %9, %4\n" "v_bfi_b32 %5, %8, %9, %5\n" "v_bfi_b32 %6, %8, %9, %6\n" "v_bfi_b32 %7, %8, %9, %7\n" );
            if (KIND == 21) RUN("v_perm_b32 %0, %0, %8, %9\n" "v_perm_b32 %1, %1, %8, %9\n" "v_perm_b32 %2, %2, %8, %9\n" "v_perm_b32 %3, %3, %8, %9\n" "v_perm_b32 %4, %4, %8, %9\n" "v_perm_b32 %5, %5, %8, %9\n" "v_perm_b32 %6, %6, %8, %9\n" "v_perm_b32 %7, %7, %8, %9\n" );
            if (KIND == 22) RUN("v_mul_lo_u32 %0, %0, %8\n" "v_mul_lo_u32 %1, %1, %8\n" "v_mul_lo_u32 %2, %2, %8\n" "v_mul_lo_u32 %3, %3, %8\n" "v_mul_lo_u32 %4, %4, %8\n" "v_mul_lo_u32 %5, %5, %8\n" "v_mul_lo_u32 %6, %6, %8\n" "v_mul_lo_u32 %7, %7, %8\n" );
            if (KIND == 23) RUN("v_mul_u32_u24 %0, %0, %8\n" "v_mul_u32_u24 %1, %1, %8\n" "v_mul_u32_u24 %2, %2, %8\n" "v_mul_u32_u24 %3, %3, %8\n" "v_mul_u32_u24 %4, %4, %8\n" "v_mul_u32_u24 %5, %5, %8\n" "v_mul_u32_u24 %6, %6, %8\n" "v_mul_u32_u24 %7, %7, %8\n" );
            if (KIND == 24) RUN("v_mad_u32_u24 %0, %0, %8, %9\n" "v_mad_u32_u24 %1, %1, %8, %9\n" "v_mad_u32_u24 %2, %2, %8, %9\n" "v_mad_u32_u24 %3, %3, %8, %9\n" "v_mad_u32_u24 %4, %4, %8, %9\n" "v_mad_u32_u24 %5, %5, %8, %9\n" "v_mad_u32_u24 %6, %6, %8, %9\n" "v_mad_u32_u24 %7, %7, %8, %9\n" );
            if (KIND == 25) RUN("v_cvt_f32_u32 %0, %0\n" "v_cvt_f32_u32 %1, %1\n" "v_cvt_f32_u32 %2, %2\n" "v_cvt_f32_u32 %3, %3\n" "v_cvt_f32_u32 %4, %4\n" "v_cvt_f32_u32 %5, %5\n" "v_cvt_f32_u32 %6, %6\n" "v_cvt_f32_u32 %7, %7\n" );
            if (KIND == 26) RUN("v_cvt_i32_f32 %0, %0\n" "v_cvt_i32_f32 %1, %1\n" "v_cvt_i32_f32 %2, %2\n" "v_cvt_i32_f32 %3, %3\n" "v_cvt_i32_f32 %4, %4\n" "v_cvt_i32_f32 %5, %5\n" "v_cvt_i32_f32 %6, %6\n" "v_cvt_i32_f32 %7, %7\n" );
            if (KIND == 27) RUN("v_floor_f32 %0, %0\n" "v_floor_f32 %1, %1\n" "v_floor_f32 %2, %2\n" "v_floor_f32 %3, %3\n" "v_floor_f32 %4, %4\n" "v_floor_f32 %5, %5\n" "v_floor_f32 %6, %6\n" "v_floor_f32 %7, %7\n" );
            if (KIND == 28) RUN("v_rcp_f32 %0, %0\n" "v_rcp_f32 %1, %1\n" "v_rcp_f32 %2, %2\n" "v_rcp_f32 %3, %3\n" "v_rcp_f32 %4, %4\n" "v_rcp_f32 %5, %5\n" "v_rcp_f32 %6, %6\n" "v_rcp_f32 %7, %7\n" );
            if (KIND == 29) RUN("v_mov_b32 %0, %8\n" "v_mov_b32 %1, %8\n" "v_mov_b32 %2, %8\n" "v_mov_b32 %3, %8\n" "v_mov_b32 %4, %8\n" "v_mov_b32 %5, %8\n" "v_mov_b32 %6, %8\n" "v_mov_b32 %7, %8\n" );
            if (KIND == 30) RUN("v_fma_f32 %0, %0, s20, %9\n" "v_fma_f32 %1, %1, s20, %9\n" "v_fma_f32 %2, %2, s20, %9\n" "v_fma_f32 %3, %3, s20, %9\n" "v_fma_f32 %4, %4, s20, %9\n" "v_fma_f32 %5, %5, s20, %9\n" "v_fma_f32 %6, %6, s20, %9\n" "v_fma_f32 %7, %7, s20, %9\n" );
            if (KIND == 31) RUN("v_div_scale_f32 %0, vcc, %0, %8, %0\n" "v_div_scale_f32 %1, vcc, %1, %8, %1\n" "v_div_scale_f32 %2, vcc, %2, %8, %2\n" "v_div_scale_f32 %3, vcc, %3, %8, %3\n" "v_div_scale_f32 %4, vcc, %4, %8, %4\n" "v_div_scale_f32 %5, vcc, %5, %8, %5\n" "v_div_scale_f32 %6, vcc, %6, %8, %6\n" "v_div_scale_f32 %7, vcc, %7, %8, %7\n" );
            if (KIND == 32) RUN("v_div_fmas_f32 %0, %0, %8, %9\n" "v_div_fmas_f32 %1, %1, %8, %9\n" "v_div_fmas_f32 %2, %2, %8, %9\n" "v_div_fmas_f32 %3, %3, %8, %9\n" "v_div_fmas_f32 %4, %4, %8, %9\n" "v_div_fmas_f32 %5, %5, %8, %9\n" "v_div_fmas_f32 %6, %6, %8, %9\n" "v_div_fmas_f32 %7, %7, %8, %9\n" );
            if (KIND == 33) RUN("v_div_fixup_f32 %0, %0, %8, %9\n" "v_div_fixup_f32 %1, %1, %8, %9\n" "v_div_fixup_f32 %2, %2, %8, %9\n" "v_div_fixup_f32 %3, %3, %8, %9\n" "v_div_fixup_f32 %4, %4, %8, %9\n" "v_div_fixup_f32 %5, %5, %8, %9\n" "v_div_fixup_f32 %6, %6, %8, %9\n" "v_div_fixup_f32 %7, %7, %8, %9\n" );
            if (KIND == 34) RUN("v_cmp_class_f32 vcc, %0, %8\n" "v_cmp_class_f32 vcc, %1, %8\n" "v_cmp_class_f32 vcc, %2, %8\n" "v_cmp_class_f32 vcc, %3, %8\n" "v_cmp_class_f32 vcc, %4, %8\n" "v_cmp_class_f32 vcc, %5, %8\n" "v_cmp_class_f32 vcc, %6, %8\n" "v_cmp_class_f32 vcc, %7, %8\n" );
            if (KIND == 35) RUN("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" );
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

typedef void (*kern_t)(float *, int, float, float);
int main()
{
    const char *names[] = { "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_max_f32", "v_min_f32", "v_max3_f32", "v_cndmask_b32", "v_cmp_lt_f32 (vcc)", "v_cmp_lt_f32 (sgpr)", "v_add_u32", "v_sub_u32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_lshl_or_b32", "v_lshl_add_u32", "v_and_or_b32", "v_bfi_b32", "v_perm_b32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_cvt_f32_u32", "v_cvt_i32_f32", "v_floor_f32", "v_rcp_f32", "v_mov_b32", "v_fma_f32 (sgpr src)", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_cmp_class_f32", "v_add_u32 sdwa" };
    kern_t ks[] = { k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>, k<18>, k<19>, k<20>, k<21>, k<22>, k<23>, k<24>, k<25>, k<26>, k<27>, k<28>, k<29>, k<30>, k<31>, k<32>, k<33>, k<34>, k<35> };
    const int nk = 36;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    float *out;
    if (hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const int wlist[] = { 1, 2, 5 };
    for (int kind = 0; kind < nk; kind++) {
        printf("%-22s", names[kind]);
        for (int wi = 0; wi < 3; wi++) {
            const int blocks = cus * 4 * wlist[wi];
            hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, 10, 1.0001f, 0.5f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 64.0 * wlist[wi];
            printf("  %d waves/SIMD: %5.2f cyc", wlist[wi], ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
