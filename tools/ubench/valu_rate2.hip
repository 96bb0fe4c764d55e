// VALU issue-rate probe (second set) for gfx950: candidates for rewriting parts of render_kernel.  8 independent chains per
// lane, W waves per SIMD; prints shader cycles per wave64 instruction per SIMD (nominal 2.4 GHz clock).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate2.hip -o tools/ubench/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define OP_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define OP_PKFMA16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define OP_PKMIN16(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
#define OP_PKMAX16(i) "v_pk_max_f16 %" #i ", %" #i ", %8\n"
#define OP_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_FMADPP(i) "v_fmac_f32_dpp %" #i ", %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_BPERM(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\n"
#define OP_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b)
{
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = (float)threadIdx.x + j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define RUN(OP) asm volatile(R8(OP) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc")
            if (KIND == 0) RUN(OP_FMA);
            if (KIND == 1) RUN(OP_MULHI);
            if (KIND == 2) RUN(OP_PKFMA16);
            if (KIND == 3) RUN(OP_PKMIN16);
            if (KIND == 4) RUN(OP_PKMAX16);
            if (KIND == 5) RUN(OP_MED3);
            if (KIND == 6) RUN(OP_MIN3);
            if (KIND == 7) RUN(OP_DPP);
            if (KIND == 8) RUN(OP_FMADPP);
            if (KIND == 9) { RUN(OP_BPERM); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            if (KIND == 10) RUN(OP_AND);
            if (KIND == 11) RUN(OP_ADD3);
            if (KIND == 12) RUN(OP_BFE);
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// 64-bit destination: v_mad_u64_u32 (both halves of a 32 x 32 product in one instruction: Philox's mulhi + mullo)
__global__ __launch_bounds__(64) void kmad(float *out, int iters, float a, float b)
{
    unsigned long long y[8];
    for (int j = 0; j < 8; j++) y[j] = threadIdx.x + j;
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                         : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(ua), "v"(ub) : "vcc");
    }
    unsigned long long s = 0;
    for (int j = 0; j < 8; j++) s += y[j];
    out[blockIdx.x * 64 + threadIdx.x] = (float)s;
}

typedef void (*kern_t)(float *, int, float, float);
int main()
{
    const char *names[] = { "v_fma_f32", "v_mul_hi_u32", "v_pk_fma_f16", "v_pk_min_f16", "v_pk_max_f16", "v_med3_f32", "v_min3_f32", "v_mov_b32 dpp", "v_fmac_f32 dpp",
                            "ds_bpermute_b32", "v_and_b32", "v_add3_u32", "v_bfe_u32", "v_mad_u64_u32" };
    kern_t ks[] = { k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, kmad };
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    float *out;
    hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const int wlist[] = { 1, 2, 5 };
    for (int kind = 0; kind < 14; kind++) {
        printf("%-18s", names[kind]);
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
