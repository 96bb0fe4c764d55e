// crt_kernels.h -- launch interface between the C ABI (crt_api.cpp) and the gfx950 kernels (crt_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "crt_traverse.h"

namespace crt {

enum { RA_ACCUM = 1u, RA_COMPACT = 2u };
// how a pixel's samples become work units (RenderArgs.chunk_mode)
enum { CM_WHOLE = 0u, CM_SLAB = 1u, CM_RING = 2u };
enum { RK_LANES = 0u, RK_POOL = 1u }; // render_kernel (pixel per lane) / render_pool_kernel (context pool in LDS)

struct RenderArgs {
    uint32_t width, height, tiles_x, tiles_y;
    uint32_t spp, sample_offset, max_depth, ids_sample;
    uint32_t tile_rank, tile_ranks, flags;
    uint32_t key0, key1;
    uint32_t local_pixels;    // 64 * (tiles of this rank): pixel slots of the work queue (set by launch_render)
    uint32_t sched_threshold; // RK_LANES: lanes that must wait for the shade/regenerate phase before it runs (1..64)
                              // RK_POOL: free lanes that end a run of walk steps (1..63)
    uint32_t chunks;          // n (a divisor of spp): a pixel is the sum, in chunk order from 0, of its n chunk sums (spp / n samples each, in sample
                              // order from 0).  0 or 1: the reference's sequential sum; so is n = spp
    uint32_t chunk_mode;      // CM_WHOLE (n <= 1): the work unit is a pixel, a lane sums its samples.  CM_SLAB: the work unit is a chunk, a lane sums
                              // its samples and stores the raw sum to `out` = the chunk scratch [pixel][chunk][3]; launch_finalize_chunks adds the
                              // chunk sums.  CM_RING: the work unit is ONE sample, the wave that rendered a block of pixels adds it up itself
                              // through its ring (crt_kernels.hip: finalize_ring_blocks).  n > 1 with CM_WHOLE: launch_render takes CM_RING
    uint32_t tiles_x_magic;   // floor(2^32 / tiles_x) for the kernel's tile -> (column, row) division (set by launch_render)
    // the work queue, set by launch_render (plan_blocks): a block is 2^group_shift pixel slots = block_units work units (pixels, or pixels x spp samples), handed to ONE wave
    uint32_t chunk_spp;       // spp / n
    uint32_t per_pixel;       // work units per pixel: 1, n (slabs) or spp (ring)
    uint32_t per_pixel_magic; // ceil(2^32 / per_pixel): k / per_pixel for k < 65536
    uint32_t group_shift, block_units, nblocks;
    // sample chunks (n >= 2): a wave adds up the chunk sums of its blocks itself, through a ring of ring_slots blocks of
    // block_units x 3 floats per persistent wave in device memory (crt_kernels.hip: finalize_ring_block)
    uint32_t ring_slots;      // in: blocks a wave may have in flight (2..16); 0 = default
    uint32_t ring_batch;      // in: complete blocks a wave lets collect before it adds them up in one pass (the pass costs the same for 1 or 10); 0 = default
    uint32_t ring_wave_base;  // index of this launch's first wave among the frame's persistent waves (the second launch of a split frame continues the first's)
    uint32_t ring_lo, ring_hi; // in: device address of the ring (render_ring_bytes() bytes); two words so that the struct stays an array of dwords to the compiler
    uint32_t blocks_per_stripe; // in: non-zero = each block counter owns a contiguous band of the frame; launch_render sets the count
    uint32_t node_cache;      // LDS copy of the top of the tree: 0 never (single waves), 1 chosen by the frame (launch_render), 2 four-wave workgroups always,
                              // 3 one 16-wave workgroup per CU with the whole tree (small trees), 4 the 16 + 4 wave split whenever it can be launched
    uint32_t kernel;          // RK_*
    uint32_t bvh_depth;       // levels of the triangle BVH (selects the LDS stack size)
    uint32_t guard_passes;    // passes of the main loop a wave may make before it gives up (set by launch_render, scaled with the frame)
    uint32_t fetch_units;     // in: work units a wave takes from a block counter at a time (rounded to whole groups of pixels); 0 = launch_render chooses
};

// counters: 32 x uint64 on the device: [0] rays [1] nodes [2] triangle tests [3] sphere tests [4] tie refolds [5] shaded hits [6] pixel work counter
// (pool kernel) [7] guard trips [8..17] wave-level schedule counters (crt_stats.sched) [18..27] section clocks (CRT_PRINT_SECTIONS) [28] block counters used up
#define CRT_COUNTER_USED_UP 28
// [7]: waves that left a render kernel through its trip-counter guard (must be 0; crt_render reports CRT_ERR_HIP otherwise)
#define CRT_COUNTER_GUARD 7
// [7] continued: a pass of render_kernel's main loop is a shade phase or a run of walk steps (0.3 us at the very least).  The limit
// (RenderArgs.guard_passes) is scaled with the frame -- 64 x 64 x a wave's fair share of the frame's path segments, 2^24..2^32 - 1
// -- so that it can only fire on a scheduling bug, not on a long frame.  crt_render reports it when it reads the counters
// (stats != NULL); the asynchronous paths (stats == NULL, crt_render_multi without stats, crt_render_noalloc) do not look.
// bytes of the counter buffer: 32 x uint64 (the above, padded) + 64 block counters of the work queue, 256 B apart
#define CRT_COUNTER_BYTES (32 * 8 + 64 * 256)
// crt_render_noalloc's scratch: a static __device__ array in crt_kernels.hip, one copy per device that loads the code
// object, holding the counter block and, behind it, the sample-chunk ring of one frame.
#define CRT_NOALLOC_SLAB_OFFSET 32768u                 // counters (CRT_COUNTER_BYTES = 16640) padded to a multiple of 256
#ifndef CRT_NOALLOC_SLAB_BYTES
#define CRT_NOALLOC_SLAB_BYTES (32u << 20)             // 32 MiB for the sample-chunk ring: persistent waves x ring slots x block units x 12 bytes must fit (include/crt.h)
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
bool pool_kernel_built(); // render_pool_kernel (RK_POOL) is only compiled with -DCRT_WITH_POOL_KERNEL
hipError_t launch_untile(const float *compact_all, uint32_t width, uint32_t height, uint32_t tile_ranks, float *out, hipStream_t stream);
hipError_t launch_finalize_chunks(const float *slabs, uint32_t chunks, size_t n, uint32_t spp, bool raw, float *out, hipStream_t stream); // CM_SLAB: slabs [pixel][chunk][3], n = 3 x pixels
// Bytes of ring a frame with these arguments needs on a device that runs max_waves persistent waves (0: whole pixels, no ring),
// and the number of ring slots that fit `budget` bytes (0: not even two).
#define CRT_RING_SLOTS_DEFAULT 4u
#define CRT_RING_BATCH_DEFAULT 1u
size_t render_ring_bytes(const RenderArgs &ra, uint32_t max_waves, uint32_t slots);
uint32_t render_ring_slots_that_fit(const RenderArgs &ra, uint32_t max_waves, size_t budget);
// The dynamic-LDS allowance of the two-launch frame's kernels on the current device (crt_kernels.hip: launch_split): called at commit.
hipError_t prepare_split_kernels();
hipError_t launch_sum_frames(const float *frames, uint32_t count, size_t n, uint32_t spp, bool raw, float *out, hipStream_t stream);      // frames n floats apart
hipError_t launch_quantize(const float *rgb, size_t n, uint8_t *out, hipStream_t stream);
hipError_t launch_finalize(const float *accum, size_t n, uint32_t total_spp, float *out, hipStream_t stream);

} // namespace crt
