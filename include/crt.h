/*
 * crt.h -- C ABI of libcrt_hip.so: the MI355X (gfx950) path-tracing hot path behind the C++ surface of
 * blueWhale903/CudaRayTracer (Scene / Renderer / Camera / LoadModel).
 *
 * Plain C: opaque handles, plain pointers and sizes, int return codes (0 = ok, non-zero = error, text via
 * crt_last_error()).  Nothing here calls exit(): the reference's checkCudaErrors (utility.h:31-40) is
 * process-fatal, this library reports and returns.  A handle is not thread-safe; different handles are.
 * Every "reference:" note names the interface of /root/reference/CudaRayTracer/ that the entry point replaces.
 */
#ifndef CRT_H
#define CRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_VERSION 200

/* ---- error convention -------------------------------------------------------------------------------- */
enum {
    CRT_OK = 0,
    CRT_ERR_INVALID = 1,   /* bad argument / bad handle state          */
    CRT_ERR_IO = 2,        /* file could not be read / parsed          */
    CRT_ERR_HIP = 3,       /* a HIP runtime call failed                */
    CRT_ERR_LIMIT = 4,     /* a documented size limit was exceeded     */
    CRT_ERR_NO_DEVICE = 5  /* no gfx950 device / kernels not loadable  */
};
/* reference: check_cuda() message (utility.h:32-40); here a thread-local string, never exit(99). */
const char *crt_last_error(void);
int crt_version(void);

/* ---- model loading ------------------------------------------------------------------------------------ */
/* reference: bool LoadModel(path, std::vector<Vertex>&, std::vector<uint32_t>&) (model_loader.h:15-55),
 * struct Vertex { vec3 position; vec3 normal; } (model_loader.h:10-13).  Own OBJ reader (v / vn / f, faces
 * fan-triangulated, vertices joined on identical (v, vn) pairs in first-use order like
 * aiProcess_JoinIdenticalVertices) instead of Assimp.  Outputs are malloc'ed; release with crt_free().
 *   *xyz_nrm : nverts * 6 floats (position.xyz, normal.xyz) -- the Vertex array
 *   *indices : ntris * 3 uint32 */
int crt_load_obj(const char *path, float **xyz_nrm, uint32_t *nverts, uint32_t **indices, uint32_t *ntris);
void crt_free(void *p);
/* SURVEY "next" row N4: the same reader, also returning which part of the file every triangle belongs to.  A group is a
 * run of faces under one `o`/`g` object and one `usemtl` material; its material record comes from the file's `mtllib`
 * (searched beside the OBJ): Kd, Ke, Ns, Ni, d, illum with the MTL defaults when a statement or the file is missing.
 * (Assimp splits such a file into several aiMesh; the reference's LoadModel concatenates their indices without the
 * base-vertex offset, model_loader.h:26-49 -- here indices always refer to the one joined vertex array.)
 *   *tri_group : ntris uint32, index into *groups;   *groups : ngroups records (malloc'ed, crt_free) */
typedef struct crt_obj_group {
    char object[64], material[64];  /* names, truncated, "" when the file has none                                 */
    float Kd[3], Ke[3];             /* diffuse colour (default 0.8), emission (default 0)                          */
    float Ns, Ni, d;                /* specular exponent (0), index of refraction (1), dissolve (1)                */
    int32_t illum;                  /* illumination model (1)                                                      */
    uint32_t first_triangle, num_triangles; /* triangles of a group are contiguous in file order                  */
} crt_obj_group;
int crt_load_obj_ex(const char *path, float **xyz_nrm, uint32_t *nverts, uint32_t **indices, uint32_t *ntris, uint32_t **tri_group,
                    crt_obj_group **groups, uint32_t *ngroups);

/* ---- scene -------------------------------------------------------------------------------------------- */
/* reference: class Scene (scene.h:20-51) whose launch_*_scene() writes a device HittableList into
 * Hittable** d_world (scene.h:158-177, 237-308); here an opaque handle that owns host staging + device SoA. */
typedef struct crt_scene crt_scene;

enum { CRT_TEX_SOLID = 0, CRT_TEX_CHECKER = 1 };                                   /* texture.h:12-44 */
enum { CRT_MAT_LAMBERTIAN = 0, CRT_MAT_METAL = 1, CRT_MAT_DIFFUSE_LIGHT = 2, CRT_MAT_DIELECTRIC = 3 }; /* material.h */

crt_scene *crt_scene_create(void);
void crt_scene_destroy(crt_scene *s); /* reference: free_world<<<1,1>>> (kernel.cu:25-28) */

/* reference: SolidColor(albedo) (texture.h:14), CheckerTexture(scale, c1, c2) (texture.h:28-29).
 * CHECKER: `a` is the reference's first ("even") colour, returned where floor-sum is ODD (texture.h:36-38). */
int crt_scene_add_texture(crt_scene *s, int type, const float a[3], const float b[3], float scale, uint32_t *out_id);
/* reference: Lambertian(color|Texture*) (material.h:22-23), Metal(color, fuzz) (material.h:42),
 * DiffuseLight(color|Texture*) (material.h:106-107), Dielectric(ior) (material.h:59).
 * texture_id = UINT32_MAX: build a SolidColor from rgb.  param = fuzz (METAL) / refraction index (DIELECTRIC). */
int crt_scene_add_material(crt_scene *s, int type, const float rgb[3], float param, uint32_t texture_id, uint32_t *out_id);
/* reference: load_triangles<<<>>> (scene.h:92-155): one Triangle per index triple, Triangle::init() box.
 * xyz has `stride_floats` floats per vertex (3 for packed positions, 6 for the Vertex array above). */
int crt_scene_add_mesh(crt_scene *s, const float *xyz, uint32_t stride_floats, uint32_t nverts, const uint32_t *indices,
                       uint32_t ntris, uint32_t material_id);
/* reference: new Sphere(center, radius, material) (scene.h:59-82,164-165; sphere.h:10-16). */
int crt_scene_add_sphere(crt_scene *s, const float center[3], float radius, uint32_t material_id);
/* reference: kernel_triangles_scene + assign_morton_codes + thrust::sort_by_key + build_bvh (scene.h:280-307).
 * Orders primitives like the reference (all triangles, then all spheres), computes each primitive's rank in
 * the reference's stable Morton order (that order decides ties, hittable_list.h:47-59), builds this library's
 * own SAH BVH (64-byte two-child nodes) and uploads everything to HIP device `device`. */
#define CRT_DEVICE_NONE (-1) /* prepare on the host only (ranks, BVH, crt_scene_get_*); crt_render then fails with CRT_ERR_NO_DEVICE */
int crt_scene_commit(crt_scene *s, int device);
/* The same with a choice of builder.  CRT_BUILD_GPU_LBVH (SURVEY "next" row N1; reference: assign_morton_codes +
 * thrust::sort_by_key + build_bvh<<<1,1>>>, bvh.h:13-185, scene.h:288-307): boxes, Morton codes, the stable radix sort
 * (= the ranks) and a Karras LBVH over that order are computed on the device in milliseconds; the tree traverses slower
 * than the host's SAH tree, the rendered result is the same.  Needs a device, triangles and at most 8 spheres; falls
 * back to the host builder otherwise (crt_scene_info.builder tells). */
enum { CRT_BUILD_GPU_LBVH = 1u,
/* CRT_BUILD_GPU_SAH (round 3; same reference loop, bvh.h:91-185): the ranks as above, and the tree by the host builder's own
 * algorithm run on the device -- 16 centroid bins per axis, area x count, leaves where a split does not pay -- level by level
 * for nodes of more than 64 triangles and one wave per smaller node (crt_gsah.hip).  The host tree's quality (same SAH cost
 * within a percent, crt_scene_info.sah_cost) in tens of milliseconds instead of seconds for a million triangles.
 * CRT_BUILDER=gpusah selects it for crt_scene_commit.  Same conditions and fall-back as CRT_BUILD_GPU_LBVH. */
       CRT_BUILD_GPU_SAH = 2u,
/* CRT_BUILD_AUTO: the library chooses -- the device's SAH builder for scenes of at least 32768 triangles (where the host takes
 * longer than the device), the host's otherwise; both give the same tree quality.  This is what crt_scene_commit(s, d) passes
 * unless the environment says CRT_BUILDER=host | gpusah | lbvh.  flags = 0 stays "host", as before. */
       CRT_BUILD_AUTO = 4u };
int crt_scene_commit_ex(crt_scene *s, int device, uint32_t flags);
/* Multi-GPU, one process driving n HIP devices (new: the reference is single-GPU, renderer.h:64-73 / kernel.cu:77-88;
 * BASELINE.json asks for tile / sample partitioning over the 8 GPUs of a node).  Like crt_scene_commit_ex, then the scene
 * is replicated: devices[k] holds the copy that logical rank k of crt_render_multi renders from; rank 0 (devices[0]) is
 * the root where frames are assembled, and the device crt_render uses.  The same device may be listed more than once
 * (logical ranks that share a GPU: how the multi-rank path is exercised on a one-GPU machine); ranks then exchange
 * their parts by device-to-device copies, otherwise through RCCL (one communicator per rank, ncclCommInitAll;
 * CRT_MULTI_TRANSPORT=copy forces the copies).  n <= 64. */
int crt_scene_commit_devices(crt_scene *s, const int *devices, uint32_t n, uint32_t flags);
/* The library reads its tuning knobs from the environment (CRT_KERNEL, CRT_SCHED_THRESHOLD, CRT_POOL_REFILL,
 * CRT_NODE_CACHE, CRT_STRIPE_BANDS, CRT_WAVES_PER_CU, CRT_PRINT_SECTIONS, CRT_CHUNK_MODE, CRT_SLAB_BUDGET_MB, CRT_RING_SLOTS,
 * CRT_RING_BATCH, CRT_FETCH_BLOCK; CRT_BVH_WIDE and CRT_MAX_LEAF shape the tree and only count at commit) once, at commit, not per frame; this reads
 * them again for a committed scene (tuning runs and tests that switch kernel variants between frames). */
int crt_scene_reload_tuning(crt_scene *s);

typedef struct crt_scene_info {
    uint32_t num_triangles, num_spheres, num_materials, num_textures;
    uint32_t num_nodes;      /* 64-byte two-child nodes or 128-byte four-child nodes (node_width) */
    uint32_t bvh_depth;      /* deepest leaf (root = depth 1), in nodes of node_width children    */
    uint32_t max_leaf_size;
    uint32_t device;         /* UINT32_MAX when prepared with CRT_DEVICE_NONE */
    uint64_t node_bytes, triangle_bytes, sphere_bytes; /* resident HBM footprint */
    float bounds_min[3], bounds_max[3];                /* triangle BVH root box  */
    uint32_t builder;        /* 0 = host SAH, 1 = GPU LBVH, 2 = GPU SAH                                      */
    float build_ms;          /* GPU builders: hipEvent time of the build kernels and the sorts; host SAH: wall time */
    uint32_t node_width;     /* 2; 4 when the scene was committed with CRT_BVH_WIDE=1 in the environment (four-child nodes: an option, slower) */
    uint32_t binary_depth;   /* depth of the two-child tree the builder made (= bvh_depth when node_width is 2)                          */
    float sah_cost;          /* surface-area cost of the two-child tree: expected node visits + 1.3 x triangle tests of a ray that hits the root box */
} crt_scene_info;
int crt_scene_get_info(const crt_scene *s, crt_scene_info *info);
/* rank[i] = position of primitive i (triangles first, then spheres) in the reference's sorted Morton order
 * (bvh.h:35-48 + scene.h:299-301).  n must equal num_triangles + num_spheres. */
int crt_scene_get_ranks(const crt_scene *s, uint32_t *rank, uint32_t n);

/* ---- camera ------------------------------------------------------------------------------------------- */
/* reference: class Camera (camera.h:11-58, 100-109).  POD image of the constructed object, passed to the
 * kernel by value.  Constructor quirks are kept: `aspect` is ignored (height = width / (16/9)), the vfov /
 * vup / defocus_angle parameters shadow the members (camera.h:13,25-28,55). */
typedef struct crt_camera {
    uint32_t image_width, image_height;
    float defocus_angle, focus_distance;
    float center[3], pixel00[3], pixel_delta_u[3], pixel_delta_v[3];
    float u[3], v[3], w[3], defocus_disk_u[3], defocus_disk_v[3], background[3];
} crt_camera;
int crt_camera_init(crt_camera *cam, uint32_t width, const float lookfrom[3], const float lookat[3], const float vup[3],
                    float vfov, float aspect, float defocus_angle, float focus_dist);

/* ---- render ------------------------------------------------------------------------------------------- */
enum {
    CRT_OUT_ACCUM = 1u,     /* out_rgb receives raw per-pixel sums (no /spp, no gamma): sample-split mode   */
    CRT_OUT_COMPACT = 2u,   /* out_rgb is this rank's tiles only, tile-major (see crt_untile): tile mode    */
    CRT_COUNT_WORK = 4u,    /* also count node fetches / primitive tests (instrumented kernel, slower)      */
    CRT_OUT_DEVICE = 8u     /* the caller states that out_rgb / out_ids are device pointers: no pointer query per frame */
};
#define CRT_TILE 8u         /* tiles are 8x8 pixels = one wave64 (reference block shape, kernel.cu:50-51)   */

typedef struct crt_render_params {
    uint32_t width, height;       /* frame size (reference: Renderer ctor, renderer.h:48)                   */
    uint32_t spp;                 /* samples rendered by this call                                          */
    uint32_t sample_offset;       /* index of the first sample (Philox counter word 1)                      */
    uint32_t max_depth;           /* reference literal 10 (camera.h:66)                                     */
    uint32_t tile_rank, tile_ranks; /* render 8x8 tiles k (row-major) with k % tile_ranks == tile_rank      */
    uint32_t flags;
    uint32_t ids_sample;          /* which sample's primary-hit ids go to out_ids                           */
    uint32_t sample_chunks;       /* 0 or 1: a lane renders all spp samples of its pixel, summed in sample order
                                     (renderer.h:28-32), bit for bit.  n > 1 (any divisor of spp): the work unit is (pixel, chunk of
                                     spp/n consecutive samples), each chunk summed in sample order, the n chunk sums added in
                                     chunk order starting from 0: same samples, finer load balance.  n = spp (one sample per
                                     unit, whatever spp is: 10, 12, 100 ...) adds single samples in sample order, i.e. it is the
                                     sequential sum again, bit for bit -- the fast exact setting; 1 < n < spp differs from it by
                                     float association only.  Where the chunk sums wait is the library's choice per frame: slabs
                                     [pixel][chunk][3] added up by a second kernel while they fit CRT_SLAB_BUDGET_MB (2048), else
                                     per-wave rings in which the wave that rendered a block of pixels adds it up itself (bounded
                                     memory whatever the frame: 4K x 1024 spp works); CRT_CHUNK_MODE=1 / 2 forces one.  Same bits. */
    uint64_t seed;                /* Philox key; reference seed literal 1984 (renderer.h:45)                */
    void *stream;                 /* hipStream_t, NULL = default stream                                     */
} crt_render_params;

typedef struct crt_stats {
    uint64_t rays;          /* path segments traced = world->hit() calls (camera.h:68)                      */
    uint64_t nodes;         /* nodes fetched (64-byte; 128-byte with CRT_BVH_WIDE=1)  (CRT_COUNT_WORK)      */
    uint64_t tri_tests;     /* 48-byte triangle records tested  (CRT_COUNT_WORK)                            */
    uint64_t sphere_tests;  /* 48-byte sphere records tested    (CRT_COUNT_WORK)                            */
    uint64_t tie_refolds;   /* rays that took the exact reference-order tie fold                            */
    uint64_t shaded_hits;   /* rays that hit something and were shaded (CRT_COUNT_WORK)                     */
    uint64_t sched[10];     /* schedule counters (CRT_COUNT_WORK): [0] shade-phase runs [1] lanes in them [2] longest wave
                               lifetime, 10 ns ticks [3] lane-passes spent waiting for a ring slot (ring mode) [4] node steps
                               [5] lanes in them [6] leaf steps [7] lanes in them
                               [8] waves that rendered pixels [9] sum of their lifetimes, 10 ns ticks               */
    float kernel_ms;        /* hipEvent time of the render kernel alone (not the chunk-sum pass) on `stream` */
    uint32_t reserved;
} crt_stats;

/* reference: Renderer::launch_kernel_render(vec3* framebuffer, Camera**, Hittable**) -> kernel_render
 * (renderer.h:18-38, 64-73).  out_rgb: width*height*3 floats row-major, row 0 = top (or the compact tile list
 * with CRT_OUT_COMPACT); out_ids (nullable): width*height int32 primary-hit primitive ids of sample
 * `ids_sample` (-1 = miss).  Both may be device pointers (any HIP allocation on the scene's device) or host
 * pointers (staged through an internal device buffer).  With stats != NULL the call synchronises `stream`
 * before returning; with stats == NULL and device pointers it is asynchronous on `stream`.
 * (A render wave that left through its trip-counter guard -- a scheduling bug, never seen -- is reported as CRT_ERR_HIP by the
 * calls that read the counters back, i.e. with stats != NULL; asynchronous calls and crt_render_noalloc cannot report it.)
 * Calls on one stream run in order.  Calls on different streams may overlap on the GPU, and may come from different host
 * threads (one thread per stream; everything else about a handle stays single-threaded): the scene keeps one set of
 * scratch buffers (work counters, chunk slabs, the staging of host-pointer frames) per stream, up to four sets; a fifth
 * stream takes over the least recently used set and waits for that set's last render.  The render kernel is persistent and fills the GPU, so what overlaps is
 * the end of one frame with the start of the next (dist.py and bench.py keep two frames in flight this way). */
int crt_render(const crt_scene *s, const crt_camera *cam, const crt_render_params *p, float *out_rgb, int32_t *out_ids,
               crt_stats *stats);

/* crt_render for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  (No counterpart in the reference, whose Renderer allocates and frees per frame, renderer.h:55-72.)
 * Same frame, same kernels, same bits as crt_render, but the work counters and the sample-chunk scratch live in a static
 * __device__ array of the library (16.6 KB of counters + CRT_NOALLOC_SLAB_BYTES = 32 MiB for slabs or rings, one copy per
 * GPU the library runs on, held until the process ends) whose per-device address is taken when a scene is committed to the
 * device.  The call makes no hipMalloc / hipFree / stream or event creation and never waits on the host: everything is
 * enqueued on p->stream (a large frame runs as two launches, the second on a helper stream of the scene that is forked from
 * and joined to p->stream by two events; stream and events are made at commit).  out_rgb and out_ids (nullable) must be device pointers; out_rays (nullable, device pointer) receives the
 * frame's path-segment count.  Limits: with sample_chunks > 1 the frame takes slabs when output pixels (rounded up to a
 * multiple of 64) x 3 x sample_chunks x 4 bytes fit the 32 MiB, else per-wave rings: (persistent waves + 16) x ring slots
 * (4, lowered to 2 if need be) x block x 12 bytes must fit, a block being the samples of whole pixels, 65 to 128 of them (spp itself
 * from 128 on) -- on a 256-CU device that admits
 * any frame size up to 256 samples per pixel (1920 x 1080 x 64 with sample_chunks = 64: 31.6 MB); CRT_ERR_LIMIT otherwise.
 * sample_chunks <= 1 needs no scratch and has no limit.  There is no
 * crt_stats and no CRT_COUNT_WORK readout.  Every call on a GPU shares that GPU's array: the call clears what must start
 * at zero, and two calls whose kernels overlap on one GPU (different streams) race -- keep them on one stream. */
#define CRT_NOALLOC_SLAB_BYTES (32u << 20)
int crt_render_noalloc(const crt_scene *s, const crt_camera *cam, const crt_render_params *p, float *out_rgb, int32_t *out_ids,
                       uint64_t *out_rays);

/* The same frame rendered by all the devices of crt_scene_commit_devices (n logical ranks), assembled on devices[0].
 *   CRT_MULTI_TILE   : 8x8 tile k goes to rank k % n (fine interleave: pixels over the mesh cost ~10x background pixels);
 *                      every rank renders a compact tile list, ONE all-gather (ncclAllGather in a group over the n
 *                      communicators, or n peer copies) and crt_untile's kernel on the root.  Bitwise equal to crt_render.
 *   CRT_MULTI_SAMPLE : rank r renders samples [r*spp/n, (r+1)*spp/n) of every pixel into raw float sums, ONE reduce(sum)
 *                      to the root (ncclReduce in a group; with peer copies the n sums are added in rank order, which is
 *                      deterministic) and crt_finalize's kernel.  Equal to crt_render up to float summation order.
 *                      spp must be a multiple of n.
 * p->tile_rank / tile_ranks must be 0 / 1 and CRT_OUT_ACCUM / CRT_OUT_COMPACT must not be set (the library partitions);
 * p->spp is the whole frame's sample count.  out_rgb: device pointer on devices[0], or host pointer.  out_ids
 * (nullable, device pointer on devices[0] or host): every rank writes the ids of its own pixels / sample straight into
 * the root's buffer, which needs peer access from its device to devices[0].
 * Consecutive frames alternate between two sets of streams and buffers per rank, so a rank's next frame starts while the
 * exchange of its previous one is still in flight.  When every rank has a device of its own, each rank's part of a frame is
 * enqueued by a worker thread of that rank (made at commit; CRT_MULTI_THREADS=0 / 1 forbids / forces), so the call costs the
 * host one rank's enqueue time plus the exchange, not n ranks'.  With stats == NULL, device pointers and CRT_OUT_DEVICE the call is
 * asynchronous: the frame is complete for work enqueued afterwards on p->stream (a stream of devices[0]; NULL = its
 * default stream).  Otherwise the call returns when the frame is in out_rgb.  stats: sums over the ranks, kernel_ms =
 * the slowest rank's kernel. */
enum { CRT_MULTI_TILE = 0, CRT_MULTI_SAMPLE = 1 };
int crt_render_multi(const crt_scene *s, const crt_camera *cam, const crt_render_params *p, int mode, float *out_rgb, int32_t *out_ids,
                     crt_stats *stats);

/* number of floats of a CRT_OUT_COMPACT buffer for one rank: ceil(ntiles / tile_ranks) * 64 * 3 (same for every rank) */
uint64_t crt_compact_floats(uint32_t width, uint32_t height, uint32_t tile_ranks);
/* Tile mode gather epilogue: `compact_all` holds the tile_ranks compact buffers back to back (rank-major, as an
 * all-gather leaves them); writes the row-major frame.  Device pointers, asynchronous on `stream`.
 * (new; the reference is single-GPU.) */
int crt_untile(const float *compact_all, uint32_t width, uint32_t height, uint32_t tile_ranks, float *out_rgb, int device, void *stream);
/* Sample mode reduce epilogue: accum (sum over `total_spp` samples) -> accum/total_spp -> gamma
 * (renderer.h:36-37, color.h:5-24).  Device pointers, in place allowed, asynchronous on `stream`. */
int crt_finalize(const float *accum, uint32_t width, uint32_t height, uint32_t total_spp, float *out_rgb, int device, void *stream);

/* ---- output path (SURVEY "next" row N2; reference: export_framebuffer_to_png, utility.h:122-145) ------------------------ */
/* Float frame -> 8-bit RGB with the reference's quantisation: (unsigned char) min(max(x * 255.0f, 0.0f), 255.0f)
 * (utility.h:133-137; truncation, NaN -> 0).  On the device (HIP kernel) when `device >= 0`: rgb and out_rgb8 are device
 * pointers, asynchronous on `stream`.  There is no host path: device < 0 is an error. */
int crt_quantize_rgb8(const float *rgb, uint32_t width, uint32_t height, uint8_t *out_rgb8, int device, void *stream);
/* 8-bit RGB rows (host memory, row 0 = top, stride = 3 * width) -> PNG file: colour type 2, bit depth 8, filter 0,
 * stored deflate blocks.  Replaces stb_image_write's stbi_write_png (utility.h:140); decoders return the same pixels. */
int crt_write_png(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* CRT_H */
