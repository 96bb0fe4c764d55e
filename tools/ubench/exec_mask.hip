// Does gfx950 skip the parts of a wave64 VALU instruction whose lanes are all inactive?  For v_fma_f32 (fast class) and v_min_f32 /
// v_cndmask_b32 (slow class): cycles per instruction per SIMD at 5 waves per SIMD with EXEC = all 64 lanes, the low 32, the low
// 16, the high 32, every other lane, and 28 scattered lanes.  Build: hipcc --offload-arch=gfx950 -O2 exec_mask.hip -o exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(OP)                                                                                                        \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                          \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])         \
                 : "v"(a), "v"(b)                                                                                         \
                 : "vcc")
#define OP_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define OP_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b, unsigned long long mask)
{
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = (float)threadIdx.x + j;
    if ((mask >> threadIdx.x) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (KIND == 0) CHAIN8(OP_FMA);
                if (KIND == 1) CHAIN8(OP_MIN);
                if (KIND == 2) CHAIN8(OP_CND);
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = p.multiProcessorCount, wps = 5, blocks = cus * 4 * wps, iters = 4000;
    float *out;
    hipMalloc(&out, sizeof(float) * 64 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = { "v_fma_f32", "v_min_f32", "v_cndmask_b32" };
    const char *mnames[] = { "all 64", "low 32", "low 16", "high 32", "even lanes", "28 scattered", "lanes 0-27" };
    const unsigned long long masks[] = { ~0ull, 0xFFFFFFFFull, 0xFFFFull, 0xFFFFFFFF00000000ull, 0x5555555555555555ull, 0x9249A4924D249249ull & 0xFFFFFFFFFFFFFFFFull, (1ull << 28) - 1 };
    void (*ks[])(float *, int, float, float, unsigned long long) = { k<0>, k<1>, k<2> };
    for (int kind = 0; kind < 3; kind++) {
        for (int m = 0; m < 7; m++) {
            hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, 10, 1.0001f, 0.5f, masks[m]);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(ks[kind], dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f, masks[m]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 64 * wps; // 8 x 8 per iteration, wps waves on a SIMD
            printf("%-14s %-13s (%2d lanes) %.2f cycles per instruction per SIMD at 2.4 GHz\n", names[kind], mnames[m], __builtin_popcountll(masks[m]), ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
    }
    return 0;
}
