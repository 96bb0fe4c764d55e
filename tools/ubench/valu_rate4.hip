// VALU issue-rate probe (fourth set: operand kinds) for gfx950: which instructions run at the 2-cycle rate of the fp32 FMA pipe and which at
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
            if (KIND == 0) RUN("v_xor_b32 %0, %0, %8\n" "v_xor_b32 %1, %1, %8\n" "v_xor_b32 %2, %2, %8\n" "v_xor_b32 %3, %3, %8\n" "v_xor_b32 %4, %4, %8\n" "v_xor_b32 %5, %5, %8\n" "v_xor_b32 %6, %6, %8\n" "v_xor_b32 %7, %7, %8\n" );
            if (KIND == 1) RUN("v_xor_b32 %0, s20, %0\n" "v_xor_b32 %1, s20, %1\n" "v_xor_b32 %2, s20, %2\n" "v_xor_b32 %3, s20, %3\n" "v_xor_b32 %4, s20, %4\n" "v_xor_b32 %5, s20, %5\n" "v_xor_b32 %6, s20, %6\n" "v_xor_b32 %7, s20, %7\n" );
            if (KIND == 2) RUN("v_xor_b32 %0, 0x9e3779b9, %0\n" "v_xor_b32 %1, 0x9e3779b9, %1\n" "v_xor_b32 %2, 0x9e3779b9, %2\n" "v_xor_b32 %3, 0x9e3779b9, %3\n" "v_xor_b32 %4, 0x9e3779b9, %4\n" "v_xor_b32 %5, 0x9e3779b9, %5\n" "v_xor_b32 %6, 0x9e3779b9, %6\n" "v_xor_b32 %7, 0x9e3779b9, %7\n" );
            if (KIND == 3) RUN("v_xor_b32 %0, 16, %0\n" "v_xor_b32 %1, 16, %1\n" "v_xor_b32 %2, 16, %2\n" "v_xor_b32 %3, 16, %3\n" "v_xor_b32 %4, 16, %4\n" "v_xor_b32 %5, 16, %5\n" "v_xor_b32 %6, 16, %6\n" "v_xor_b32 %7, 16, %7\n" );
            if (KIND == 4) RUN("v_add_f32 %0, s20, %0\n" "v_add_f32 %1, s20, %1\n" "v_add_f32 %2, s20, %2\n" "v_add_f32 %3, s20, %3\n" "v_add_f32 %4, s20, %4\n" "v_add_f32 %5, s20, %5\n" "v_add_f32 %6, s20, %6\n" "v_add_f32 %7, s20, %7\n" );
            if (KIND == 5) RUN("v_add_f32 %0, 0x33d6bf95, %0\n" "v_add_f32 %1, 0x33d6bf95, %1\n" "v_add_f32 %2, 0x33d6bf95, %2\n" "v_add_f32 %3, 0x33d6bf95, %3\n" "v_add_f32 %4, 0x33d6bf95, %4\n" "v_add_f32 %5, 0x33d6bf95, %5\n" "v_add_f32 %6, 0x33d6bf95, %6\n" "v_add_f32 %7, 0x33d6bf95, %7\n" );
            if (KIND == 6) RUN("v_add_f32 %0, 1.0, %0\n" "v_add_f32 %1, 1.0, %1\n" "v_add_f32 %2, 1.0, %2\n" "v_add_f32 %3, 1.0, %3\n" "v_add_f32 %4, 1.0, %4\n" "v_add_f32 %5, 1.0, %5\n" "v_add_f32 %6, 1.0, %6\n" "v_add_f32 %7, 1.0, %7\n" );
            if (KIND == 7) RUN("v_mul_f32 %0, s20, %0\n" "v_mul_f32 %1, s20, %1\n" "v_mul_f32 %2, s20, %2\n" "v_mul_f32 %3, s20, %3\n" "v_mul_f32 %4, s20, %4\n" "v_mul_f32 %5, s20, %5\n" "v_mul_f32 %6, s20, %6\n" "v_mul_f32 %7, s20, %7\n" );
            if (KIND == 8) RUN("v_mul_f32 %0, 0x37800000, %0\n" "v_mul_f32 %1, 0x37800000, %1\n" "v_mul_f32 %2, 0x37800000, %2\n" "v_mul_f32 %3, 0x37800000, %3\n" "v_mul_f32 %4, 0x37800000, %4\n" "v_mul_f32 %5, 0x37800000, %5\n" "v_mul_f32 %6, 0x37800000, %6\n" "v_mul_f32 %7, 0x37800000, %7\n" );
            if (KIND == 9) RUN("v_add_u32 %0, s20, %0\n" "v_add_u32 %1, s20, %1\n" "v_add_u32 %2, s20, %2\n" "v_add_u32 %3, s20, %3\n" "v_add_u32 %4, s20, %4\n" "v_add_u32 %5, s20, %5\n" "v_add_u32 %6, s20, %6\n" "v_add_u32 %7, s20, %7\n" );
            if (KIND == 10) RUN("v_add_u32 %0, 0xffffff00, %0\n" "v_add_u32 %1, 0xffffff00, %1\n" "v_add_u32 %2, 0xffffff00, %2\n" "v_add_u32 %3, 0xffffff00, %3\n" "v_add_u32 %4, 0xffffff00, %4\n" "v_add_u32 %5, 0xffffff00, %5\n" "v_add_u32 %6, 0xffffff00, %6\n" "v_add_u32 %7, 0xffffff00, %7\n" );
            if (KIND == 11) RUN("v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_fma_f32 %3, %3, %8, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_fma_f32 %6, %6, %8, %9\n" "v_fma_f32 %7, %7, %8, %9\n" );
            if (KIND == 12) RUN("v_fma_f32 %0, %0, %8, -%9\n" "v_fma_f32 %1, %1, %8, -%9\n" "v_fma_f32 %2, %2, %8, -%9\n" "v_fma_f32 %3, %3, %8, -%9\n" "v_fma_f32 %4, %4, %8, -%9\n" "v_fma_f32 %5, %5, %8, -%9\n" "v_fma_f32 %6, %6, %8, -%9\n" "v_fma_f32 %7, %7, %8, -%9\n" );
            if (KIND == 13) RUN("v_fma_f32 %0, %0, s20, %9\n" "v_fma_f32 %1, %1, s20, %9\n" "v_fma_f32 %2, %2, s20, %9\n" "v_fma_f32 %3, %3, s20, %9\n" "v_fma_f32 %4, %4, s20, %9\n" "v_fma_f32 %5, %5, s20, %9\n" "v_fma_f32 %6, %6, s20, %9\n" "v_fma_f32 %7, %7, s20, %9\n" );
            if (KIND == 14) RUN("v_fma_f32 %0, %0, %8, 1.0\n" "v_fma_f32 %1, %1, %8, 1.0\n" "v_fma_f32 %2, %2, %8, 1.0\n" "v_fma_f32 %3, %3, %8, 1.0\n" "v_fma_f32 %4, %4, %8, 1.0\n" "v_fma_f32 %5, %5, %8, 1.0\n" "v_fma_f32 %6, %6, %8, 1.0\n" "v_fma_f32 %7, %7, %8, 1.0\n" );
            if (KIND == 15) RUN("v_fmac_f32 %0, %8, %9\n" "v_fmac_f32 %1, %8, %9\n" "v_fmac_f32 %2, %8, %9\n" "v_fmac_f32 %3, %8, %9\n" "v_fmac_f32 %4, %8, %9\n" "v_fmac_f32 %5, %8, %9\n" "v_fmac_f32 %6, %8, %9\n" "v_fmac_f32 %7, %8, %9\n" );
            if (KIND == 16) RUN("v_cndmask_b32_e32 %0, %0, %8, vcc\n" "v_cndmask_b32_e32 %1, %1, %8, vcc\n" "v_cndmask_b32_e32 %2, %2, %8, vcc\n" "v_cndmask_b32_e32 %3, %3, %8, vcc\n" "v_cndmask_b32_e32 %4, %4, %8, vcc\n" "v_cndmask_b32_e32 %5, %5, %8, vcc\n" "v_cndmask_b32_e32 %6, %6, %8, vcc\n" "v_cndmask_b32_e32 %7, %7, %8, vcc\n" );
            if (KIND == 17) RUN("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n" "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n" "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n" "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n" "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n" "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n" "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n" );
            if (KIND == 18) RUN("v_cndmask_b32_e64 %0, %0, %8, vcc\n" "v_cndmask_b32_e64 %1, %1, %8, vcc\n" "v_cndmask_b32_e64 %2, %2, %8, vcc\n" "v_cndmask_b32_e64 %3, %3, %8, vcc\n" "v_cndmask_b32_e64 %4, %4, %8, vcc\n" "v_cndmask_b32_e64 %5, %5, %8, vcc\n" "v_cndmask_b32_e64 %6, %6, %8, vcc\n" "v_cndmask_b32_e64 %7, %7, %8, vcc\n" );
            if (KIND == 19) RUN("v_mov_b32 %0, s20\n" "v_mov_b32 %1, s20\n" "v_mov_b32 %2, s20\n" "v_mov_b32 %3, s20\n" "v_mov_b32 %4, s20\n" "v_mov_b32 %5, s20\n" "v_mov_b32 %6, s20\n" "v_mov_b32 %7, s20\n" );
            if (KIND == 20) RUN("v_mov_b32 %0, 0x7149f2ca\n" "v_mov_b32 %1, 0x7149f2ca\n" "v_mov_b32 %2, 0x7149f2ca\n" "v_mov_b32 %3, 0x7149f2ca\n" "v_mov_b32 %4, 0x7149f2ca\n" "v_mov_b32 %5, 0x7149f2ca\n" "v_mov_b32 %6, 0x7149f2ca\n" "v_mov_b32 %7, 0x7149f2ca\n" );
            if (KIND == 21) RUN("v_lshlrev_b32 %0, 4, %0\n" "v_lshlrev_b32 %1, 4, %1\n" "v_lshlrev_b32 %2, 4, %2\n" "v_lshlrev_b32 %3, 4, %3\n" "v_lshlrev_b32 %4, 4, %4\n" "v_lshlrev_b32 %5, 4, %5\n" "v_lshlrev_b32 %6, 4, %6\n" "v_lshlrev_b32 %7, 4, %7\n" );
            if (KIND == 22) RUN("v_add_u32 %0, %0, %0\n" "v_add_u32 %1, %1, %1\n" "v_add_u32 %2, %2, %2\n" "v_add_u32 %3, %3, %3\n" "v_add_u32 %4, %4, %4\n" "v_add_u32 %5, %5, %5\n" "v_add_u32 %6, %6, %6\n" "v_add_u32 %7, %7, %7\n" );
            if (KIND == 23) RUN("v_and_b32 %0, s20, %0\n" "v_and_b32 %1, s20, %1\n" "v_and_b32 %2, s20, %2\n" "v_and_b32 %3, s20, %3\n" "v_and_b32 %4, s20, %4\n" "v_and_b32 %5, s20, %5\n" "v_and_b32 %6, s20, %6\n" "v_and_b32 %7, s20, %7\n" );
            if (KIND == 24) RUN("v_sub_f32 %0, s20, %0\n" "v_sub_f32 %1, s20, %1\n" "v_sub_f32 %2, s20, %2\n" "v_sub_f32 %3, s20, %3\n" "v_sub_f32 %4, s20, %4\n" "v_sub_f32 %5, s20, %5\n" "v_sub_f32 %6, s20, %6\n" "v_sub_f32 %7, s20, %7\n" );
            if (KIND == 25) RUN("v_max_f32 %0, s20, %0\n" "v_max_f32 %1, s20, %1\n" "v_max_f32 %2, s20, %2\n" "v_max_f32 %3, s20, %3\n" "v_max_f32 %4, s20, %4\n" "v_max_f32 %5, s20, %5\n" "v_max_f32 %6, s20, %6\n" "v_max_f32 %7, s20, %7\n" );
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

typedef void (*kern_t)(float *, int, float, float);
int main()
{
    const char *names[] = { "v_xor_b32 vgpr", "v_xor_b32 sgpr", "v_xor_b32 literal", "v_xor_b32 inline", "v_add_f32 sgpr", "v_add_f32 literal", "v_add_f32 inline", "v_mul_f32 sgpr", "v_mul_f32 literal", "v_add_u32 sgpr", "v_add_u32 literal", "v_fma_f32 vgpr", "v_fma_f32 neg mods", "v_fma_f32 sgpr", "v_fma_f32 inline 1.0", "v_fmac_f32 e32", "v_cndmask e32 vcc", "v_cndmask e64 sgpr", "v_cndmask e64 vcc", "v_mov_b32 sgpr", "v_mov_b32 literal", "v_lshlrev_b32 by 4", "v_mul_u32 by add", "v_and_b32 sgpr", "v_sub_f32 sgpr", "v_max_f32 sgpr" };
    kern_t ks[] = { k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>, k<18>, k<19>, k<20>, k<21>, k<22>, k<23>, k<24>, k<25> };
    const int nk = 26;
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
