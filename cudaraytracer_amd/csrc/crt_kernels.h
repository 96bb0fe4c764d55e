// crt_kernels.h -- launch interface between the C ABI (crt_api.cpp) and the gfx950 kernels (crt_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "crt_traverse.h"

namespace crt {

enum { RA_ACCUM = 1u, RA_COMPACT = 2u };
enum { RK_LANES = 0u, RK_POOL = 1u }; // render_kernel (pixel per lane) / render_pool_kernel (context pool in LDS)

struct RenderArgs {
    uint32_t width, height, tiles_x, tiles_y;
    uint32_t spp, sample_offset, max_depth, ids_sample;
    uint32_t tile_rank, tile_ranks, flags;
    uint32_t key0, key1;
    uint32_t local_pixels;    // 64 * (tiles of this rank): pixel slots of the work queue (set by launch_render)
    uint32_t sched_threshold; // RK_LANES: lanes that must wait for the shade/regenerate phase before it runs (1..64)
                              // RK_POOL: free lanes that end a run of walk steps (1..63)
    uint32_t chunks;          // sample chunks per pixel (a power of two dividing spp): work unit = (pixel, chunk)
    uint32_t chunk_shift;     // log2(chunks)                                  (set by launch_render)
    uint32_t tiles_x_magic;   // floor(2^32 / tiles_x) for the kernel's tile -> (column, row) division (set by launch_render)
    uint32_t blocks_per_stripe; // in: non-zero = each block counter owns a contiguous band of the frame; launch_render sets the count
    uint32_t node_cache;      // LDS copy of the top of the tree: 0 never (single waves), 1 chosen by the frame (launch_render), 2 four-wave workgroups always,
                              // 3 one 16-wave workgroup per CU with the whole tree (small trees), 4 the 16 + 4 wave split whenever it can be launched
    uint32_t kernel;          // RK_*
    uint32_t bvh_depth;       // levels of the triangle BVH (selects the LDS stack size)
    uint32_t fetch_shift;     // log2 of the work units a wave takes from a block counter at a time; in: 0 = launch_render chooses
};

// counters: 32 x uint64 on the device: [0] rays [1] nodes [2] triangle tests [3] sphere tests [4] tie refolds [5] shaded hits [6] pixel work counter
// (pool kernel) [7] guard trips [8..17] wave-level schedule counters (crt_stats.sched) [18..26] section clocks (CRT_PRINT_SECTIONS) [28] block counters used up
#define CRT_COUNTER_USED_UP 28
// [7]: waves that left a render kernel through its trip-counter guard (must be 0; crt_render reports CRT_ERR_HIP otherwise)
#define CRT_COUNTER_GUARD 7
// passes of render_kernel's main loop a wave may make (a pass is a shade phase or a run of walk steps, 0.3 us at the very
// least: 2^28 passes are minutes, three orders of magnitude above the longest frame in BASELINE.json)
#define CRT_GUARD_PASSES (1u << 28)
// bytes of the counter buffer: 32 x uint64 (the above, padded) + 64 block counters of the work queue, 256 B apart
#define CRT_COUNTER_BYTES (32 * 8 + 64 * 256)
// crt_render_noalloc's scratch: a static __device__ array in crt_kernels.hip, one copy per device that loads the code
// object, holding the counter block and, behind it, the sample-chunk slabs of one frame.
#define CRT_NOALLOC_SLAB_OFFSET 32768u                 // counters (CRT_COUNTER_BYTES = 16640) padded to a multiple of 256
#ifndef CRT_NOALLOC_SLAB_BYTES
#define CRT_NOALLOC_SLAB_BYTES (32u << 20)             // 32 MiB of chunk slabs: output floats x sample_chunks x 4 must fit (include/crt.h)
#endif
// Takes the address of `device`'s copy of the array (hipGetSymbolAddress; `device` must be the current device) and keeps
// it; called once per device when a scene is committed to it.  noalloc_scratch_ptr returns the kept address (nullptr
// before that).
hipError_t noalloc_scratch_init(int device);
void *noalloc_scratch_ptr(int device);
// A second stream and two events of the caller's (created ahead of time, on the device of `stream`): with them launch_render may
// split a frame over two concurrent launches that share the frame's work queue (crt_kernels.hip: launch_split).  nullptr: one launch.
struct SplitStreams { hipStream_t helper; hipEvent_t fork, join; };
hipError_t launch_render(const SceneView &sc, const CameraPOD &cam, const RenderArgs &ra, float *out, int *ids, unsigned long long *counters,
                         bool count_work, uint32_t max_waves, hipStream_t stream, const SplitStreams *split = nullptr);
hipError_t launch_untile(const float *compact_all, uint32_t width, uint32_t height, uint32_t tile_ranks, float *out, hipStream_t stream);
hipError_t launch_finalize_chunks(const float *slabs, uint32_t chunks, size_t n, uint32_t spp, bool raw, float *out, hipStream_t stream); // slabs [64-pixel block][chunk][64][3], n = 3 x pixels
// The dynamic-LDS allowance of the two-launch frame's kernels on the current device (crt_kernels.hip: launch_split): called at commit.
hipError_t prepare_split_kernels();
hipError_t launch_sum_frames(const float *frames, uint32_t count, size_t n, uint32_t spp, bool raw, float *out, hipStream_t stream);      // frames n floats apart
hipError_t launch_quantize(const float *rgb, size_t n, uint8_t *out, hipStream_t stream);
hipError_t launch_finalize(const float *accum, size_t n, uint32_t total_spp, float *out, hipStream_t stream);

} // namespace crt
