// Does v_cndmask_b32 in its VOP2 form (implicit VCC) issue slower than the VOP3 form on gfx950, and does it depend on who wrote
// VCC last?  8 independent chains, 5 waves per SIMD; cycles per wave64 instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/cndmask_vcc.hip -o tools/ubench/cndmask_vcc
#include <hip/hip_runtime.h>
#include <cstdio>
#define C8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define E32(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define E64V(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define E64S(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b)
{
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = (float)threadIdx.x + j;
#define ARGS : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc", "s20", "s21"
    if (KIND == 1 || KIND == 4) asm volatile("v_cmp_lt_f32 vcc, %8, %9\n" ARGS);   // VCC written by the VALU once, before the loop
    if (KIND == 2) asm volatile("s_mov_b64 vcc, -1\ns_mov_b64 s[20:21], -1\n" ARGS); // by the SALU
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0 || KIND == 1 || KIND == 2) asm volatile(C8(E32) ARGS);
            if (KIND == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %9\n" C8(E32) ARGS);               // one compare, eight selects (9 instructions)
            if (KIND == 4) asm volatile(C8(E64V) ARGS);
            if (KIND == 5) asm volatile("v_cmp_lt_f32 vcc, %0, %9\ns_and_b64 vcc, vcc, exec\n" C8(E32) ARGS); // VALU compare, SALU and, selects
            if (KIND == 6) asm volatile("v_cmp_lt_f32 s[20:21], %0, %9\n" C8(E64S) ARGS);
            if (KIND == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %9\nv_cndmask_b32_e32 %0, %0, %8, vcc\nv_cmp_lt_f32 vcc, %1, %9\nv_cndmask_b32_e32 %1, %1, %8, vcc\n"
                                        "v_cmp_lt_f32 vcc, %2, %9\nv_cndmask_b32_e32 %2, %2, %8, vcc\nv_cmp_lt_f32 vcc, %3, %9\nv_cndmask_b32_e32 %3, %3, %8, vcc\n" ARGS); // 8 instructions
            if (KIND == 8) asm volatile("v_cmp_lt_f32 s[20:21], %0, %9\nv_cndmask_b32_e64 %0, %0, %8, s[20:21]\nv_cmp_lt_f32 s[20:21], %1, %9\nv_cndmask_b32_e64 %1, %1, %8, s[20:21]\n"
                                        "v_cmp_lt_f32 s[20:21], %2, %9\nv_cndmask_b32_e64 %2, %2, %8, s[20:21]\nv_cmp_lt_f32 s[20:21], %3, %9\nv_cndmask_b32_e64 %3, %3, %8, s[20:21]\n" ARGS);
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
typedef void (*kern_t)(float *, int, float, float);
int main()
{
    const char *names[] = { "e32 vcc (vcc never written)", "e32 vcc (VALU wrote vcc once)", "e32 vcc (SALU wrote vcc once)", "v_cmp + 8 x e32 (per 9 instr)", "e64 vcc",
                            "v_cmp, s_and, 8 x e32 (per 8)", "v_cmp sgpr + 8 x e64 (per 9)", "4 x (v_cmp vcc, e32) (per 8)", "4 x (v_cmp sgpr, e64) (per 8)" };
    const double per[] = { 8, 8, 8, 9, 8, 8, 9, 8, 8 };
    kern_t ks[] = { k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8> };
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    float *out;
    if (hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, waves = 5;
    for (int kind = 0; kind < 9; kind++) {
        const int blocks = cus * 4 * waves;
        hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, 10, 1.0001f, 0.5f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %5.2f cycles per instruction\n", names[kind], ms * 1e-3 * 2.4e9 / ((double)iters * 8.0 * per[kind] * waves));
    }
    return 0;
}
