// crt_kernels.hip -- gfx950 kernels of the hot path and their launchers.
//
// render_kernel: one wave64 = one 8x8 pixel tile (the reference's block shape, kernel.cu:50-51, renderer.h:51-52),
// one lane = one pixel, all samples of the pixel summed in the lane in sample order (renderer.h:28-32), so the
// float sum is the reference's sum whatever the tile-to-GPU assignment.  The traversal stack is an LDS column per
// lane (stack[level][lane]: bank = lane, conflict free).
#include <hip/hip_runtime.h>

#include "crt_kernels.h"
#include "crt_traverse.h"

namespace crt {

struct LdsStack {
    int *col; // &lds[lane], stride 64 ints
    int sp;
    __device__ __forceinline__ void reset() { sp = 0; }
    __device__ __forceinline__ void push(int v) { col[sp * 64] = v; sp++; }
    __device__ __forceinline__ int pop()
    {
        if (sp == 0) return CRT_BVH_NONE;
        sp--;
        return col[sp * 64];
    }
};

struct RegCounters {
    uint32_t nodes = 0, tris = 0, spheres = 0, rays = 0, refolds = 0, shades = 0;
    __device__ __forceinline__ void node() { nodes++; }
    __device__ __forceinline__ void tri() { tris++; }
    __device__ __forceinline__ void sphere() { spheres++; }
    __device__ __forceinline__ void ray() { rays++; }
    __device__ __forceinline__ void refold() { refolds++; }
    __device__ __forceinline__ void shade() { shades++; }
};
struct RayCounters { // always-on: rays and tie refolds only
    uint32_t rays = 0, refolds = 0;
    __device__ __forceinline__ void node() {}
    __device__ __forceinline__ void tri() {}
    __device__ __forceinline__ void sphere() {}
    __device__ __forceinline__ void ray() { rays++; }
    __device__ __forceinline__ void refold() { refolds++; }
    __device__ __forceinline__ void shade() {}
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool COUNT_WORK>
__global__ __launch_bounds__(64) void render_kernel(SceneView sc, CameraPOD cam, RenderArgs ra, float *__restrict__ out,
                                                    int *__restrict__ ids, unsigned long long *__restrict__ counters)
{
    __shared__ int lds_stack[CRT_STACK_DEPTH * 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t tile = blockIdx.x * ra.tile_ranks + ra.tile_rank;
    if (tile >= ra.tiles_x * ra.tiles_y) return;
    const uint32_t tx = tile % ra.tiles_x, ty = tile / ra.tiles_x;
    const uint32_t i = tx * 8u + (lane & 7u), j = ty * 8u + (lane >> 3);
    const bool inside = i < ra.width && j < ra.height; // renderer.h:22
    const uint32_t pixel = j * ra.width + i;            // renderer.h:24

    LdsStack stack;
    stack.col = lds_stack + lane;
    stack.sp = 0;
    typename std::conditional<COUNT_WORK, RegCounters, RayCounters>::type cnt;

    f3 col = mk3(0.0f, 0.0f, 0.0f);
    int primary = -1;
    if (inside) {
        for (uint32_t s = 0; s < ra.spp; s++) { // renderer.h:29-32
            const uint32_t sample = ra.sample_offset + s;
            int prim;
            const f3 c = sample_radiance(sc, cam, i, j, pixel, sample, ra.max_depth, ra.key0, ra.key1, stack, cnt, prim);
            col = col + c;
            if (sample == ra.ids_sample) primary = prim;
        }
    }
    if (inside || (ra.flags & RA_COMPACT)) {
        f3 o = col;
        if (!(ra.flags & RA_ACCUM)) { // renderer.h:36-37, color.h:5-24
            o = div3(col, (float)ra.spp);
            o.x = o.x > 0 ? __builtin_sqrtf(o.x) : 0.0f;
            o.y = o.y > 0 ? __builtin_sqrtf(o.y) : 0.0f;
            o.z = o.z > 0 ? __builtin_sqrtf(o.z) : 0.0f;
        }
        const size_t idx = (ra.flags & RA_COMPACT) ? ((size_t)blockIdx.x * 64u + lane) : (size_t)pixel;
        out[3 * idx + 0] = o.x; out[3 * idx + 1] = o.y; out[3 * idx + 2] = o.z;
    }
    if (ids && inside && ra.ids_sample >= ra.sample_offset && ra.ids_sample - ra.sample_offset < ra.spp) ids[pixel] = primary;

    const uint32_t rays = wave_sum(cnt.rays), refolds = wave_sum(cnt.refolds);
    if (lane == 0) {
        atomicAdd(&counters[0], (unsigned long long)rays);
        if (refolds) atomicAdd(&counters[4], (unsigned long long)refolds);
    }
    if constexpr (COUNT_WORK) {
        const uint32_t n = wave_sum(cnt.nodes), t = wave_sum(cnt.tris), sp = wave_sum(cnt.spheres), sh = wave_sum(cnt.shades);
        if (lane == 0) {
            atomicAdd(&counters[5], (unsigned long long)sh);
            atomicAdd(&counters[1], (unsigned long long)n);
            atomicAdd(&counters[2], (unsigned long long)t);
            atomicAdd(&counters[3], (unsigned long long)sp);
        }
    }
}

// compact (rank-major, tile-major) -> row-major frame
__global__ __launch_bounds__(64) void untile_kernel(const float *__restrict__ compact_all, uint32_t width, uint32_t height, uint32_t tiles_x,
                                                    uint32_t ntiles, uint32_t tile_ranks, uint32_t tiles_per_rank, float *__restrict__ out)
{
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    if (tile >= ntiles) return;
    const uint32_t rank = tile % tile_ranks, slot = tile / tile_ranks;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t i = tx * 8u + (lane & 7u), j = ty * 8u + (lane >> 3);
    if (i >= width || j >= height) return;
    const size_t src = ((size_t)rank * tiles_per_rank + slot) * 64u + lane;
    const size_t dst = (size_t)j * width + i;
    out[3 * dst + 0] = compact_all[3 * src + 0];
    out[3 * dst + 1] = compact_all[3 * src + 1];
    out[3 * dst + 2] = compact_all[3 * src + 2];
}

// accum / total_spp, then gamma (renderer.h:36-37, color.h:5-24)
__global__ __launch_bounds__(256) void finalize_kernel(const float *__restrict__ accum, size_t n, float spp, float *__restrict__ out)
{
    const size_t k = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const float v = accum[k] / spp;
    out[k] = v > 0 ? __builtin_sqrtf(v) : 0.0f;
}

hipError_t launch_render(const SceneView &sc, const CameraPOD &cam, const RenderArgs &ra, float *out, int *ids, unsigned long long *counters,
                         bool count_work, hipStream_t stream)
{
    const uint32_t ntiles = ra.tiles_x * ra.tiles_y;
    const uint32_t blocks = (ntiles + ra.tile_ranks - 1 - ra.tile_rank) / ra.tile_ranks; // tiles k = b*ranks + rank < ntiles
    if (blocks == 0) return hipSuccess;
    if (count_work) hipLaunchKernelGGL(render_kernel<true>, dim3(blocks), dim3(64), 0, stream, sc, cam, ra, out, ids, counters);
    else hipLaunchKernelGGL(render_kernel<false>, dim3(blocks), dim3(64), 0, stream, sc, cam, ra, out, ids, counters);
    return hipGetLastError();
}

hipError_t launch_untile(const float *compact_all, uint32_t width, uint32_t height, uint32_t tile_ranks, float *out, hipStream_t stream)
{
    const uint32_t tiles_x = (width + 7u) / 8u, tiles_y = (height + 7u) / 8u, ntiles = tiles_x * tiles_y;
    const uint32_t tiles_per_rank = (ntiles + tile_ranks - 1) / tile_ranks;
    hipLaunchKernelGGL(untile_kernel, dim3(ntiles), dim3(64), 0, stream, compact_all, width, height, tiles_x, ntiles, tile_ranks, tiles_per_rank, out);
    return hipGetLastError();
}

hipError_t launch_finalize(const float *accum, size_t n, uint32_t total_spp, float *out, hipStream_t stream)
{
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(256), 0, stream, accum, n, (float)total_spp, out);
    return hipGetLastError();
}

} // namespace crt
