// Is v_rsq_f32 (or v_sqrt_f32) followed by fma refinement steps the correctly rounded square root?  All positive inputs on the GPU
// against sqrtf compiled with -fhip-fp32-correctly-rounded-divide-sqrt, mismatches counted for x in [2^-100, 2^100].
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/ubench/sqrt_exact.hip -o tools/ubench/sqrt_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void check_all(uint32_t hi, unsigned long long *bad)
{
    const uint32_t bits = (hi << 24) | (blockIdx.x * 256u + threadIdx.x);
    const float x = __uint_as_float(bits);
    if (!(x >= 0x1p-100f && x <= 0x1p+100f)) return;
    const float want = __builtin_sqrtf(x);
    const float r = __builtin_amdgcn_rsqf(x), h = 0.5f * r;
    const float s0 = x * r;
    const float s1 = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), h, s0);
    const float s2 = __builtin_fmaf(__builtin_fmaf(-s1, s1, x), h, s1);
    const float q = __builtin_amdgcn_sqrtf(x);
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q, q, x), h, q);
    if (__float_as_uint(s0) != __float_as_uint(want)) atomicAdd(&bad[0], 1ull);
    if (__float_as_uint(s1) != __float_as_uint(want)) atomicAdd(&bad[1], 1ull);
    if (__float_as_uint(s2) != __float_as_uint(want)) atomicAdd(&bad[2], 1ull);
    if (__float_as_uint(q) != __float_as_uint(want)) atomicAdd(&bad[3], 1ull);
    if (__float_as_uint(q1) != __float_as_uint(want)) atomicAdd(&bad[4], 1ull);
    atomicAdd(&bad[5], 1ull);
}
int main()
{
    unsigned long long *bad;
    if (hipMalloc(&bad, 64) != hipSuccess) { printf("no device\n"); return 1; }
    hipMemset(bad, 0, 64);
    for (uint32_t hi = 0; hi < 128; hi++) hipLaunchKernelGGL(check_all, dim3(65536), dim3(256), 0, 0, hi, bad);
    unsigned long long h[8];
    if (hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return 1; }
    printf("inputs checked %llu; mismatches against the correctly rounded sqrt: x*rsq(x) %llu, + one fma step %llu, + two steps %llu; v_sqrt_f32 %llu, + one step %llu\n",
           h[5], h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
