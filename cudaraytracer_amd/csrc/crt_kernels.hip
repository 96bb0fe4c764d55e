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

// Per-lane traversal stack: an LDS column (stack[level][lane]: bank = lane, conflict free).  Level 0 holds the
// CRT_BVH_NONE sentinel for the whole kernel, so pop() needs no empty check: popping it ends the walk.
struct LdsStack {
    int *top; // next free slot of this lane's column (stride 64 ints)
    int *base;
    __device__ __forceinline__ void init(int *column) { column[0] = CRT_BVH_NONE; base = column + 64; top = base; }
    __device__ __forceinline__ void reset() { top = base; }
    __device__ __forceinline__ void push(int v) { *top = v; top += 64; }
    __device__ __forceinline__ int pop() { top -= 64; return *top; }
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

// renderer.h:36-37 + color.h:5-24: average, gamma, store (or the raw sum in accumulate mode)
__device__ __forceinline__ void store_pixel(const RenderArgs &ra, float *__restrict__ out, int *__restrict__ ids, uint32_t q, uint32_t pixel, f3 col,
                                            int primary)
{
    f3 o = col;
    if (!(ra.flags & RA_ACCUM)) {
        o = div3(col, (float)ra.spp);
        o.x = o.x > 0 ? __builtin_sqrtf(o.x) : 0.0f;
        o.y = o.y > 0 ? __builtin_sqrtf(o.y) : 0.0f;
        o.z = o.z > 0 ? __builtin_sqrtf(o.z) : 0.0f;
    }
    const size_t idx = (ra.flags & RA_COMPACT) ? (size_t)q : (size_t)pixel;
    out[3 * idx + 0] = o.x; out[3 * idx + 1] = o.y; out[3 * idx + 2] = o.z;
    if (ids && ra.ids_sample >= ra.sample_offset && ra.ids_sample - ra.sample_offset < ra.spp) ids[pixel] = primary;
}

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }

// Lane phases while not walking (a lane walks the BVH iff tv.cur != CRT_BVH_NONE)
enum { P_DONE = 0, P_FETCH = 1, P_NEW = 2, P_HIT = 3 };

// render_kernel: one wave64 per workgroup, persistent.  A lane computes a pixel start to finish (its samples summed in
// sample order, renderer.h:28-32, so the float sum is the reference's whatever the partition) and takes pixels from a
// device-wide counter in 8x8-tile order (q = 64 * tile + lane-in-tile), so a lane that finishes its pixel does not wait
// for the slowest lane of its tile.
//
// Divergence control.  Paths end after 1..max_depth segments; 58 % of the segments of the headline scene never enter
// the triangle BVH (they miss its root box), the others walk 14 steps on average with a long tail.  Lock-step "sample
// loop / bounce loop / traversal loop" nesting leaves ~80 % of the lanes idle (profiles/r1_v1_nested_loops).  Here
// every lane is a small state machine
//     FETCH (needs a pixel) -> NEW (needs a camera ray) -> walking -> HIT (walk over: resolve, shade, scatter) -> ...
// and the wave alternates, by 64-bit ballots, between two phases:
//   A  "shade", run when at least `sched_threshold` lanes are not walking (or nobody walks): HIT lanes resolve their
//      hit, add emission and end the path or ask for scatter random numbers; finished pixels are stored and new ones
//      fetched; NEW lanes ask for camera random numbers; ONE Philox evaluation serves both kinds; then scatter or
//      camera ray, the sphere tests and the root-box test start the next walk.  A defocus-disk rejection
//      (utility.h:100-109) retries in the next run with the next Philox block.
//   B  "walk": one node step for the lanes at a node or one leaf step for the lanes at a leaf, whichever group is
//      larger, until enough lanes wait for phase A again.
// A lane's own operation sequence is exactly the sequential one (crt_traverse.h: sample_radiance), so results do not
// depend on the schedule.  (Two pixel contexts per lane were tried and measured slower: DESIGN.md section 7.)
#ifndef CRT_WAVES_PER_EU
#define CRT_WAVES_PER_EU 5 // 512 VGPRs / 5 waves -> 96 registers per lane; the few spills this forces sit on the rare tie path
#endif
#if CRT_WAVES_PER_EU > 0
#define CRT_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu(CRT_WAVES_PER_EU, CRT_WAVES_PER_EU)))
#else
#define CRT_OCCUPANCY_ATTR
#endif
template <bool COUNT_WORK>
__global__ __launch_bounds__(64) CRT_OCCUPANCY_ATTR void render_kernel(SceneView sc, CameraPOD cam, RenderArgs ra, float *__restrict__ out,
                                                                       int *__restrict__ ids, unsigned long long *__restrict__ counters)
{
    __shared__ int lds_stack[(CRT_STACK_DEPTH + 1) * 64];
    const uint32_t lane = threadIdx.x;
    unsigned int *work = reinterpret_cast<unsigned int *>(counters + 6);
    const uint32_t NO_PIXEL = 0xFFFFFFFFu;
    const float tmin = 0.001f; // camera.h:68

    LdsStack stack;
    stack.init(lds_stack + lane);
    typename std::conditional<COUNT_WORK, RegCounters, RayCounters>::type cnt;

    uint32_t q = NO_PIXEL, i = 0, j = 0, pixel = 0;
    f3 col = mk3(0.0f, 0.0f, 0.0f);
    int primary = -1, path_prim = -1;
    int phase = P_FETCH;
    uint32_t s_idx = 0, depth = 0;
    Path p;
    p.o = p.d = p.attenuation = p.color = mk3(0.0f, 0.0f, 0.0f);
    Search hit;
    hit.t = FLT_MAX; hit.code = -1; hit.mat = 0; hit.rank = 0; hit.near_tie = false;
    Trav tv;
    tv.idir = tv.ood = mk3(0.0f, 0.0f, 0.0f); tv.shift = 0.0f; tv.hi = 0.0f; tv.cur = CRT_BVH_NONE;
    const int threshold = (int)ra.sched_threshold;
    uint32_t wc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; // schedule counters, lane 0 only (COUNT_WORK)
    unsigned long long t_start = 0;
    if constexpr (COUNT_WORK) t_start = wall_clock64(); // 100 MHz

    for (;;) {
        const bool walking = tv.cur != CRT_BVH_NONE;
        const int nwalk = popc64(__ballot(walking));
        const int nready = popc64(__ballot(!walking && phase != P_DONE));
        if (nready + nwalk == 0) break;
        if (nready >= threshold || nwalk == 0) {
            // ================================ phase A ================================
            if constexpr (COUNT_WORK) { if (lane == 0) { wc[0]++; wc[1] += nready; } }
            if (!walking && phase != P_DONE) {
                bool want_scatter = false;
                Surface sf;
                sf.point = sf.normal = mk3(0.0f, 0.0f, 0.0f); sf.front_face = false; sf.prim = -1;
                int m_type = MAT_DIFFUSE_LIGHT;
                float m_param = 0.0f;
                f3 colour = mk3(0.0f, 0.0f, 0.0f);
                if (phase == P_HIT) {
                    float t = 0.0f;
                    int mat = 0;
                    const int code = resolve_hit(sc, p.o, p.d, tmin, FLT_MAX, hit, t, mat, stack, cnt);
                    if (code < 0) { // camera.h:68-71
                        p.color = mk3(fma1(p.attenuation.x, cam.background.x, p.color.x), fma1(p.attenuation.y, cam.background.y, p.color.y),
                                      fma1(p.attenuation.z, cam.background.z, p.color.z));
                    } else {
                        const Shade m = sc.shades[mat]; // issued together with the primitive fetch below: one memory round trip per hit
                        sf = surface_at(sc, code, p.o, p.d, t);
                        cnt.shade();
                        if (depth == 0) path_prim = sf.prim;
                        m_type = m.type; m_param = m.param;
                        colour = texture_value(m, sf.point);
                        const f3 e = emitted(m_type, colour); // camera.h:73-74
                        p.color = mk3(fma1(p.attenuation.x, e.x, p.color.x), fma1(p.attenuation.y, e.y, p.color.y), fma1(p.attenuation.z, e.z, p.color.z));
                        // camera.h:66,78: the scatter of the last allowed bounce cannot reach the image: not evaluated
                        want_scatter = scatters(m_type) && depth + 1 < ra.max_depth;
                    }
                    if (!want_scatter) { // path over: renderer.h:31
                        col = col + p.color;
                        if (ra.sample_offset + s_idx == ra.ids_sample) primary = path_prim;
                        s_idx++;
                        phase = s_idx < ra.spp ? P_NEW : P_FETCH;
                    }
                }
                if (phase == P_FETCH) {
                    if (q != NO_PIXEL) store_pixel(ra, out, ids, q, pixel, col, primary);
                    // Raster tile order on purpose.  Measured slower: fetching one pixel ahead (+15 %: a lane then holds a
                    // pixel hostage at the end of the frame), the tiles over the mesh first (+14 %), spread evenly (+5 %),
                    // or started at the mesh's first tile row (+11 %): DESIGN.md section 7.
                    for (;;) {
                        q = atomicAdd(work, 1u);
                        if (q >= ra.local_pixels) { q = NO_PIXEL; phase = P_DONE; break; }
                        const uint32_t tile = (q >> 6) * ra.tile_ranks + ra.tile_rank, l = q & 63u;
                        i = (tile % ra.tiles_x) * 8u + (l & 7u);
                        j = (tile / ra.tiles_x) * 8u + (l >> 3);
                        if (i < ra.width && j < ra.height) { // renderer.h:22
                            pixel = j * ra.width + i;      // renderer.h:24
                            s_idx = 0;
                            col = mk3(0.0f, 0.0f, 0.0f);
                            primary = -1;
                            phase = P_NEW;
                            break;
                        }
                        if (ra.flags & RA_COMPACT) { out[3 * (size_t)q + 0] = 0.0f; out[3 * (size_t)q + 1] = 0.0f; out[3 * (size_t)q + 2] = 0.0f; }
                    }
                }
                bool has_ray = false;
                if (want_scatter) {
                    const u4 r = philox4x32_10(pixel, ra.sample_offset + s_idx, 1u + depth, 0u, ra.key0, ra.key1);
                    f3 att, no, nd;
                    scatter(m_type, m_param, colour, p.d, sf.point, sf.normal, sf.front_face, r, att, no, nd);
                    p.attenuation = p.attenuation * att; // camera.h:82-83
                    p.o = no; p.d = nd;
                    depth++;
                    has_ray = true;
                }
                if (phase == P_NEW) {
                    camera_ray(cam, i, j, pixel, ra.sample_offset + s_idx, ra.key0, ra.key1, p.o, p.d);
                    p.attenuation = mk3(1.0f, 1.0f, 1.0f);
                    p.color = mk3(0.0f, 0.0f, 0.0f);
                    depth = 0;
                    path_prim = -1;
                    has_ray = true;
                }
                if (has_ray) {
                    cnt.ray();
                    hit.t = FLT_MAX; hit.code = -1; hit.near_tie = false;
                    trav_begin<MODE_CLOSEST>(sc, p.o, p.d, tmin, hit, 0, 0.0f, tv, stack, cnt);
                    phase = P_HIT; // what the lane does once tv.cur is CRT_BVH_NONE again (at once if the root box is missed)
                }
            }
        } else {
            // ================================ phase B ================================
            const int alive = nready + nwalk;
            for (;;) {
                const bool inner = is_inner(tv.cur), leaf = tv.cur < 0;
                const int n_inner = popc64(__ballot(inner)), n_leaf = popc64(__ballot(leaf));
                const int nw = n_inner + n_leaf;
                if (nw == 0 || alive - nw >= threshold) break;
                if (n_inner >= n_leaf) {
                    if constexpr (COUNT_WORK) { if (lane == 0) { wc[4]++; wc[5] += n_inner; } }
                    if (inner) trav_inner(sc, tv, stack, cnt);
                } else {
                    if constexpr (COUNT_WORK) { if (lane == 0) { wc[6]++; wc[7] += n_leaf; } }
                    if (leaf) trav_leaf<MODE_CLOSEST>(sc, p.o, p.d, tmin, hit, 0, 0.0f, tv, stack, cnt);
                }
            }
        }
    }

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
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t w = wave_sum(wc[k]);
            if (lane == 0) atomicAdd(&counters[8 + k], (unsigned long long)w);
        }
        // load balance: waves that did work, the sum and the maximum of their lifetimes (10 ns ticks)
        if (lane == 0 && wc[0] > 1) {
            const unsigned long long dt = wall_clock64() - t_start;
            atomicAdd(&counters[16], 1ull);
            atomicAdd(&counters[17], dt);
            atomicMax(&counters[10], dt);
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
                         bool count_work, uint32_t max_waves, hipStream_t stream)
{
    const uint32_t ntiles = ra.tiles_x * ra.tiles_y;
    const uint32_t local_tiles = (ntiles + ra.tile_ranks - 1 - ra.tile_rank) / ra.tile_ranks; // tiles k = b*ranks + rank < ntiles
    if (local_tiles == 0) return hipSuccess;
    RenderArgs a = ra;
    a.local_pixels = local_tiles * 64u;
    // persistent waves: enough to fill the device; each lane keeps drawing pixels until the counter runs out
    const uint32_t blocks = local_tiles < max_waves ? local_tiles : max_waves;
    if (count_work) hipLaunchKernelGGL(render_kernel<true>, dim3(blocks), dim3(64), 0, stream, sc, cam, a, out, ids, counters);
    else hipLaunchKernelGGL(render_kernel<false>, dim3(blocks), dim3(64), 0, stream, sc, cam, a, out, ids, counters);
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
