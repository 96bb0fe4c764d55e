// VALU issue-rate probe for gfx950: for each instruction kind, 8 independent chains per lane, W waves per SIMD;
// prints shader cycles per wave64 instruction per SIMD (nominal clock).  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHAIN8(OP)                                                                                                        \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                          \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])         \
                 : "v"(a), "v"(b), "s"(sa)                                                                                 \
                 : "vcc", "scc")

#define OP_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define OP_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define OP_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define OP_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define OP_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define OP_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define OP_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_MULHI24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define OP_RDLANE(i) "v_readlane_b32 s40, %" #i ", 3\n"
#define OP_WRLANE(i) "v_writelane_b32 %" #i ", %10, 3\n"
#define OP_CVT(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define OP_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define OP_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_DIVFIXUP(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_SALU(i) "s_add_u32 s40, s40, %10\n"
#define OP_SALU64(i) "s_and_b64 s[40:41], s[40:41], exec\n"

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b, int sa)
{
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = (float)threadIdx.x + j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) CHAIN8(OP_FMA);
            if (KIND == 1) CHAIN8(OP_MUL);
            if (KIND == 2) CHAIN8(OP_ADD);
            if (KIND == 3) CHAIN8(OP_XOR);
            if (KIND == 4) CHAIN8(OP_ADDU);
            if (KIND == 5) CHAIN8(OP_MOV);
            if (KIND == 6) CHAIN8(OP_CND);
            if (KIND == 7) CHAIN8(OP_CMP);
            if (KIND == 8) CHAIN8(OP_MIN);
            if (KIND == 9) CHAIN8(OP_MAX3);
            if (KIND == 10) CHAIN8(OP_MULLO);
            if (KIND == 11) CHAIN8(OP_MULHI);
            if (KIND == 12) CHAIN8(OP_MUL24);
            if (KIND == 13) CHAIN8(OP_MULHI24);
            if (KIND == 14) CHAIN8(OP_MAD24);
            if (KIND == 15) CHAIN8(OP_RCP);
            if (KIND == 16) CHAIN8(OP_SQRT);
            if (KIND == 17) asm volatile(OP_RDLANE(0) OP_RDLANE(1) OP_RDLANE(2) OP_RDLANE(3) OP_RDLANE(4) OP_RDLANE(5) OP_RDLANE(6) OP_RDLANE(7)
                                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                                         : "v"(a), "v"(b), "s"(sa) : "s40", "scc");
            if (KIND == 18) CHAIN8(OP_WRLANE);
            if (KIND == 19) CHAIN8(OP_CVT);
            if (KIND == 20) CHAIN8(OP_LSHL);
            if (KIND == 21) CHAIN8(OP_DIVSCALE);
            if (KIND == 22) CHAIN8(OP_DIVFMAS);
            if (KIND == 23) CHAIN8(OP_DIVFIXUP);
            if (KIND == 24) asm volatile(OP_SALU(0) OP_SALU(1) OP_SALU(2) OP_SALU(3) OP_SALU(4) OP_SALU(5) OP_SALU(6) OP_SALU(7)
                                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                                         : "v"(a), "v"(b), "s"(sa) : "s40", "scc");
            if (KIND == 25) asm volatile(OP_SALU64(0) OP_SALU64(1) OP_SALU64(2) OP_SALU64(3) OP_SALU64(4) OP_SALU64(5) OP_SALU64(6) OP_SALU64(7)
                                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                                         : "v"(a), "v"(b), "s"(sa) : "s40", "s41", "scc");
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// packed fp32 (two floats per lane in a 64-bit register pair)
#define OP_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define CHAIN8PK(OP)                                                                                                            \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                              \
                 : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7])            \
                 : "v"(da), "v"(db))
template <int KIND>
__global__ __launch_bounds__(64) void kpk(float *out, int iters, float a, float b, int)
{
    double y[8];
    for (int j = 0; j < 8; j++) y[j] = __hiloint2double(__float_as_int((float)threadIdx.x + j), __float_as_int(1.0f + j));
    const double da = __hiloint2double(__float_as_int(a), __float_as_int(a)), db = __hiloint2double(__float_as_int(b), __float_as_int(b));
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) CHAIN8PK(OP_PKFMA);
            if (KIND == 1) CHAIN8PK(OP_PKMUL);
            if (KIND == 2) CHAIN8PK(OP_PKADD);
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += __int_as_float(__double2loint(y[j])) + __int_as_float(__double2hiint(y[j]));
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

typedef void (*kern_t)(float *, int, float, float, int);
template <int K> static kern_t get() { return k<K>; }

int main(int argc, char **argv)
{
    const char *names[] = { "v_fma_f32", "v_mul_f32", "v_add_f32", "v_xor_b32", "v_add_u32", "v_mov_b32", "v_cndmask_b32", "v_cmp_lt_f32", "v_min_f32",
                            "v_max3_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mad_u32_u24", "v_rcp_f32",
                            "v_sqrt_f32", "v_readlane_b32", "v_writelane_b32", "v_cvt_f32_u32", "v_lshlrev_b32", "v_div_scale_f32", "v_div_fmas_f32",
                            "v_div_fixup_f32", "s_add_u32", "s_and_b64", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32" };
    kern_t ks[] = { get<0>(), get<1>(), get<2>(), get<3>(), get<4>(), get<5>(), get<6>(), get<7>(), get<8>(), get<9>(), get<10>(), get<11>(), get<12>(),
                    get<13>(), get<14>(), get<15>(), get<16>(), get<17>(), get<18>(), get<19>(), get<20>(), get<21>(), get<22>(), get<23>(), get<24>(), get<25>(),
                    kpk<0>, kpk<1>, kpk<2> };
    const int first = argc > 1 ? atoi(argv[1]) : 0;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    float *out;
    hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const int wlist[] = { 1, 2, 5 };
    for (int kind = first; kind < 29; kind++) {
        printf("%-18s", names[kind]);
        for (int wi = 0; wi < 3; wi++) {
            const int wps = wlist[wi];
            const int blocks = cus * 4 * wps;
            hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, 50, 1.0001f, 0.5f, 1);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f, 1);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 64 * wps;
            const double cyc = ms * 1e-3 * p.clockRate * 1e3;
            printf("  %d waves/SIMD: %6.2f cyc/instr", wps, cyc / instr_per_simd);
        }
        printf("   (%.2f GHz nominal)\n", p.clockRate * 1e-6);
    }
    return 0;
}
