// Is v_rcp_f32 followed by one or two fma refinement steps the correctly rounded reciprocal?  All 2^32 inputs on the GPU against
// IEEE division 1.0f / x (compiled with -fhip-fp32-correctly-rounded-divide-sqrt), mismatches counted for |x| in [2^-100, 2^100].
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/ubench/rcp_exact.hip -o tools/ubench/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void check(unsigned long long *bad, uint32_t *examples)
{
    const uint32_t bits = blockIdx.x * 256u + threadIdx.x + blockIdx.y * 0x01000000u * 256u / 256u * 0u; // (filled below)
    (void)bits;
}
__global__ __launch_bounds__(256) void check_all(uint32_t hi, unsigned long long *bad, uint32_t *examples)
{
    const uint32_t bits = (hi << 24) | (blockIdx.x * 256u + threadIdx.x);
    const float x = __uint_as_float(bits);
    const float ax = __builtin_fabsf(x);
    if (!(ax >= 0x1p-100f && ax <= 0x1p+100f)) return;
    const float want = 1.0f / x;
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e0 = __builtin_fmaf(-x, y0, 1.0f);
    const float y1 = __builtin_fmaf(e0, y0, y0);
    const float e1 = __builtin_fmaf(-x, y1, 1.0f);
    const float y2 = __builtin_fmaf(e1, y1, y1);
    if (__float_as_uint(y0) != __float_as_uint(want)) atomicAdd(&bad[0], 1ull);
    if (__float_as_uint(y1) != __float_as_uint(want)) { if (atomicAdd(&bad[1], 1ull) < 8) examples[atomicAdd(&examples[31], 1u) & 7u] = bits; }
    if (__float_as_uint(y2) != __float_as_uint(want)) { if (atomicAdd(&bad[2], 1ull) < 8) examples[8 + (atomicAdd(&examples[30], 1u) & 7u)] = bits; }
    atomicAdd(&bad[3], 1ull);
}
int main()
{
    unsigned long long *bad; uint32_t *ex;
    if (hipMalloc(&bad, 32) != hipSuccess || hipMalloc(&ex, 128) != hipSuccess) { printf("no device\n"); return 1; }
    hipMemset(bad, 0, 32); hipMemset(ex, 0, 128);
    for (uint32_t hi = 0; hi < 256; hi++) hipLaunchKernelGGL(check_all, dim3(65536), dim3(256), 0, 0, hi, bad, ex);
    unsigned long long h[4]; uint32_t e[32];
    if (hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return 1; }
    hipMemcpy(e, ex, 128, hipMemcpyDeviceToHost);
    printf("inputs checked %llu: v_rcp_f32 alone differs from 1/x in %llu, + one fma step in %llu, + two steps in %llu\n", h[3], h[0], h[1], h[2]);
    for (int k = 0; k < 8 && k < (int)h[1]; k++) printf("  one step differs at 0x%08x\n", e[k]);
    for (int k = 0; k < 8 && k < (int)h[2]; k++) printf("  two steps differ at 0x%08x\n", e[8 + k]);
    return 0;
}
