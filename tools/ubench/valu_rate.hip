// VALU issue-rate probe: N independent fma chains per lane, W waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PK>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (PK) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            } else {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out; hipMalloc(&out, sizeof(float) * 64 * cus * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int pk = 0; pk < 2; pk++)
    for (int wps = 1; wps <= 8; wps++) {
        const int blocks = cus * 4 * wps;
        if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, 100, 1.0001f, 0.5f); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, 100, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 64 * wps;       // wave-instructions per SIMD
        const double cyc = ms * 1e-3 * p.clockRate * 1e3;             // clockRate in kHz
        printf("%s waves/SIMD %d: %.3f ms, %.2f cycles per lane-fma-instruction-equivalent (64 per iter) (at %.2f GHz nominal)\n", pk ? "compiler(pk)" : "v_fma_f32", wps, ms, cyc / instr_per_simd, p.clockRate * 1e-6);
    }
    return 0;
}
