// The select tail of render_kernel's node step as the compiler writes it (VOP2 v_cndmask reading VCC, VCC rewritten by s_and_b64)
// against the same with every mask in an ordinary SGPR pair (VOP3 forms): cycles per 13-instruction sequence per SIMD, 5 waves.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/cndmask_seq.hip -o tools/ubench/cndmask_seq
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b)
{
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = (float)threadIdx.x + j;
#define ARGS : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc", "scc", "s14", "s15", "s16", "s17", "s18", "s19"
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) asm volatile(
                "v_cmp_lt_f32_e32 vcc, %0, %1\n"
                "v_cmp_le_f32_e64 s[14:15], %0, %2\n"
                "s_nop 0\n"
                "v_cndmask_b32_e32 %3, %4, %5, vcc\n"
                "s_and_b64 vcc, s[14:15], vcc\n"
                "v_cndmask_b32_e32 %6, %5, %4, vcc\n"
                "v_cmp_le_f32_e32 vcc, %1, %2\n"
                "v_cndmask_b32_e64 %7, %7, %4, s[14:15]\n"
                "s_and_b64 s[16:17], vcc, s[14:15]\n"
                "s_or_b64 s[14:15], vcc, s[14:15]\n"
                "v_cndmask_b32_e32 %0, %7, %6, vcc\n"
                "v_cndmask_b32_e64 %1, 0, %8, s[16:17]\n"
                "v_cndmask_b32_e64 %2, %9, 0, s[14:15]\n" ARGS);
            if (KIND == 1) asm volatile(
                "v_cmp_lt_f32_e64 s[18:19], %0, %1\n"
                "v_cmp_le_f32_e64 s[14:15], %0, %2\n"
                "s_nop 0\n"
                "v_cndmask_b32_e64 %3, %4, %5, s[18:19]\n"
                "s_and_b64 s[18:19], s[14:15], s[18:19]\n"
                "v_cndmask_b32_e64 %6, %5, %4, s[18:19]\n"
                "v_cmp_le_f32_e64 s[18:19], %1, %2\n"
                "v_cndmask_b32_e64 %7, %7, %4, s[14:15]\n"
                "s_and_b64 s[16:17], s[18:19], s[14:15]\n"
                "s_or_b64 s[14:15], s[18:19], s[14:15]\n"
                "v_cndmask_b32_e64 %0, %7, %6, s[18:19]\n"
                "v_cndmask_b32_e64 %1, 0, %8, s[16:17]\n"
                "v_cndmask_b32_e64 %2, %9, 0, s[14:15]\n" ARGS);
            if (KIND == 2) asm volatile( // the VOP2 selects only, each after an unrelated VALU instruction
                "v_cmp_lt_f32_e32 vcc, %0, %1\n"
                "v_add_f32 %3, %3, %8\n"
                "v_cndmask_b32_e32 %4, %4, %5, vcc\n"
                "v_add_f32 %6, %6, %8\n"
                "v_cndmask_b32_e32 %7, %7, %5, vcc\n"
                "v_add_f32 %0, %0, %8\n"
                "v_cndmask_b32_e32 %1, %1, %5, vcc\n"
                "v_add_f32 %2, %2, %8\n" ARGS);
            if (KIND == 3) asm volatile( // the same with VOP3 selects
                "v_cmp_lt_f32_e64 s[18:19], %0, %1\n"
                "v_add_f32 %3, %3, %8\n"
                "v_cndmask_b32_e64 %4, %4, %5, s[18:19]\n"
                "v_add_f32 %6, %6, %8\n"
                "v_cndmask_b32_e64 %7, %7, %5, s[18:19]\n"
                "v_add_f32 %0, %0, %8\n"
                "v_cndmask_b32_e64 %1, %1, %5, s[18:19]\n"
                "v_add_f32 %2, %2, %8\n" ARGS);
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
typedef void (*kern_t)(float *, int, float, float);
int main()
{
    const char *names[] = { "node-step select tail, as compiled (VOP2 + VCC)", "node-step select tail, masks in SGPR pairs (VOP3)", "cmp, then 3 x (add, VOP2 select on VCC)", "cmp, then 3 x (add, VOP3 select)" };
    kern_t ks[] = { k<0>, k<1>, k<2>, k<3> };
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    float *out;
    if (hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, waves = 5;
    for (int kind = 0; kind < 4; kind++) {
        const int blocks = cus * 4 * waves;
        hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, 10, 1.0001f, 0.5f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %6.1f cycles per sequence\n", names[kind], ms * 1e-3 * 2.4e9 / ((double)iters * 8.0 * waves));
    }
    return 0;
}
