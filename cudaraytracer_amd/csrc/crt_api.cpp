// crt_api.cpp -- the C ABI declared in include/crt.h.  Host code only; all device work goes through
// crt_kernels.h.  There is no CPU rendering path in this library: without a usable HIP device every render
// entry point fails with CRT_ERR_NO_DEVICE / CRT_ERR_HIP.
#include "../../include/crt.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>
#include "crt_bvh.h"
#include "crt_kernels.h"
#include "crt_lbvh.h"
#include <chrono>
#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: the library is loaded with dlopen on first multi-device use

using namespace crt;

// ---------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------
static thread_local std::string g_error;

static int fail(int code, const std::string &msg)
{
    g_error = msg;
    return code;
}
static int hip_fail(hipError_t e, const char *what)
{
    (void)hipGetLastError();
    const int code = (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNoBinaryForGpu || e == hipErrorInvalidDeviceFunction ||
                      e == hipErrorSharedObjectInitFailed) ? CRT_ERR_NO_DEVICE : CRT_ERR_HIP;
    return fail(code, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

extern "C" const char *crt_last_error(void) { return g_error.c_str(); }
extern "C" int crt_version(void) { return CRT_VERSION; }
extern "C" void crt_free(void *p) { free(p); }

// ---------------------------------------------------------------------------------------------------------
// OBJ reader (replaces Assimp in LoadModel, model_loader.h:15-55)
// ---------------------------------------------------------------------------------------------------------
static bool parse_corner(const char *&p, long nv, long nn, long &vi, long &ni)
{
    while (*p == ' ' || *p == '\t') p++;
    if (*p == 0 || *p == '\n' || *p == '\r' || *p == '#') return false;
    char *end;
    long v = strtol(p, &end, 10);
    if (end == p) return false;
    p = end;
    long t = 0, n = 0;
    bool has_n = false;
    if (*p == '/') {
        p++;
        if (*p != '/') { t = strtol(p, &end, 10); p = end; }
        if (*p == '/') { p++; n = strtol(p, &end, 10); has_n = end != p; p = end; }
    }
    (void)t;
    vi = v > 0 ? v - 1 : nv + v;
    ni = has_n ? (n > 0 ? n - 1 : nn + n) : -1;
    return true;
}

namespace {
struct MtlRecord { std::string name; float Kd[3] = { 0.8f, 0.8f, 0.8f }, Ke[3] = { 0, 0, 0 }, Ns = 0.0f, Ni = 1.0f, d = 1.0f; int illum = 1; };

std::string first_word(const char *p)
{
    while (*p == ' ' || *p == '\t') p++;
    const char *e = p;
    while (*e && *e != '\n' && *e != '\r') e++;
    while (e > p && (e[-1] == ' ' || e[-1] == '\t')) e--;
    return std::string(p, e);
}

void read_mtl(const std::string &path, std::vector<MtlRecord> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return; // a missing library leaves the defaults (Assimp does the same with a warning)
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        const char *p = line;
        while (*p == ' ' || *p == '\t') p++;
        char *e;
        if (!strncmp(p, "newmtl", 6) && (p[6] == ' ' || p[6] == '\t')) { out.emplace_back(); out.back().name = first_word(p + 7); continue; }
        if (out.empty()) continue;
        MtlRecord &m = out.back();
        if (!strncmp(p, "Kd ", 3)) { p += 3; for (int k = 0; k < 3; k++) { m.Kd[k] = strtof(p, &e); p = e; } }
        else if (!strncmp(p, "Ke ", 3)) { p += 3; for (int k = 0; k < 3; k++) { m.Ke[k] = strtof(p, &e); p = e; } }
        else if (!strncmp(p, "Ns ", 3)) m.Ns = strtof(p + 3, &e);
        else if (!strncmp(p, "Ni ", 3)) m.Ni = strtof(p + 3, &e);
        else if (!strncmp(p, "d ", 2)) m.d = strtof(p + 2, &e);
        else if (!strncmp(p, "illum ", 6)) m.illum = (int)strtol(p + 6, &e, 10);
    }
    fclose(f);
}
} // namespace

extern "C" int crt_load_obj_ex(const char *path, float **xyz_nrm, uint32_t *nverts, uint32_t **indices, uint32_t *ntris, uint32_t **tri_group,
                               crt_obj_group **groups, uint32_t *ngroups)
{
    if (!path || !xyz_nrm || !nverts || !indices || !ntris) return fail(CRT_ERR_INVALID, "crt_load_obj: null argument");
    if ((tri_group || groups || ngroups) && !(tri_group && groups && ngroups)) return fail(CRT_ERR_INVALID, "crt_load_obj_ex: group outputs go together");
    *xyz_nrm = nullptr; *indices = nullptr; *nverts = 0; *ntris = 0;
    if (tri_group) { *tri_group = nullptr; *groups = nullptr; *ngroups = 0; }
    FILE *f = fopen(path, "rb");
    if (!f) return fail(CRT_ERR_IO, std::string("crt_load_obj: cannot open ") + path);
    std::vector<float> pos, nrm;
    std::vector<float> verts;         // 6 floats per joined vertex
    std::vector<uint32_t> idx, tgroup;
    std::vector<crt_obj_group> grp;
    std::vector<MtlRecord> mtl;
    std::string cur_object, cur_material;
    bool group_open = false;
    std::map<std::pair<long, long>, uint32_t> joined; // (v, vn) -> vertex, first use order
    std::vector<char> line(1 << 16);
    int rc = CRT_OK;
    while (fgets(line.data(), (int)line.size(), f)) {
        const char *p = line.data();
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char *e;
            p += 2;
            for (int k = 0; k < 3; k++) { pos.push_back(strtof(p, &e)); p = e; }
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            char *e;
            p += 3;
            for (int k = 0; k < 3; k++) { nrm.push_back(strtof(p, &e)); p = e; }
        } else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t')) {
            cur_object = first_word(p + 2); group_open = false;
        } else if (!strncmp(p, "usemtl", 6) && (p[6] == ' ' || p[6] == '\t')) {
            cur_material = first_word(p + 7); group_open = false;
        } else if (!strncmp(p, "mtllib", 6) && (p[6] == ' ' || p[6] == '\t')) {
            if (tri_group) {
                std::string dir(path);
                const size_t slash = dir.find_last_of('/');
                dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
                read_mtl(dir + first_word(p + 7), mtl);
            }
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            std::vector<uint32_t> corner;
            long vi, ni;
            const long nv = (long)pos.size() / 3, nn = (long)nrm.size() / 3;
            while (parse_corner(p, nv, nn, vi, ni)) {
                if (vi < 0 || vi >= nv || ni >= nn) { rc = fail(CRT_ERR_IO, std::string("crt_load_obj: face index out of range in ") + path); break; }
                auto key = std::make_pair(vi, ni);
                auto it = joined.find(key);
                if (it == joined.end()) {
                    const uint32_t id = (uint32_t)(verts.size() / 6);
                    for (int k = 0; k < 3; k++) verts.push_back(pos[3 * (size_t)vi + k]);
                    for (int k = 0; k < 3; k++) verts.push_back(ni >= 0 ? nrm[3 * (size_t)ni + k] : 0.0f);
                    it = joined.emplace(key, id).first;
                }
                corner.push_back(it->second);
            }
            if (rc != CRT_OK) break;
            if (!group_open && corner.size() >= 3) {
                crt_obj_group g;
                memset(&g, 0, sizeof g);
                snprintf(g.object, sizeof g.object, "%s", cur_object.c_str());
                snprintf(g.material, sizeof g.material, "%s", cur_material.c_str());
                g.first_triangle = (uint32_t)(idx.size() / 3);
                grp.push_back(g);
                group_open = true;
            }
            for (size_t k = 2; k < corner.size(); k++) { // aiProcess_Triangulate: fan
                idx.push_back(corner[0]); idx.push_back(corner[k - 1]); idx.push_back(corner[k]);
                tgroup.push_back((uint32_t)grp.size() - 1);
                grp.back().num_triangles++;
            }
        }
    }
    fclose(f);
    if (rc != CRT_OK) return rc;
    if (idx.empty()) return fail(CRT_ERR_IO, std::string("crt_load_obj: no faces in ") + path);
    for (crt_obj_group &g : grp) { // material records (the library may come after the faces in the file)
        MtlRecord m;
        for (const MtlRecord &r : mtl) if (r.name == g.material) m = r;
        for (int k = 0; k < 3; k++) { g.Kd[k] = m.Kd[k]; g.Ke[k] = m.Ke[k]; }
        g.Ns = m.Ns; g.Ni = m.Ni; g.d = m.d; g.illum = m.illum;
    }
    *xyz_nrm = (float *)malloc(verts.size() * sizeof(float));
    *indices = (uint32_t *)malloc(idx.size() * sizeof(uint32_t));
    bool ok = *xyz_nrm && *indices;
    if (tri_group) {
        *tri_group = (uint32_t *)malloc(tgroup.size() * sizeof(uint32_t));
        *groups = (crt_obj_group *)malloc(grp.size() * sizeof(crt_obj_group));
        ok = ok && *tri_group && *groups;
    }
    if (!ok) {
        free(*xyz_nrm); free(*indices); *xyz_nrm = nullptr; *indices = nullptr;
        if (tri_group) { free(*tri_group); free(*groups); *tri_group = nullptr; *groups = nullptr; }
        return fail(CRT_ERR_LIMIT, "crt_load_obj: out of memory");
    }
    memcpy(*xyz_nrm, verts.data(), verts.size() * sizeof(float));
    memcpy(*indices, idx.data(), idx.size() * sizeof(uint32_t));
    *nverts = (uint32_t)(verts.size() / 6);
    *ntris = (uint32_t)(idx.size() / 3);
    if (tri_group) {
        memcpy(*tri_group, tgroup.data(), tgroup.size() * sizeof(uint32_t));
        memcpy(*groups, grp.data(), grp.size() * sizeof(crt_obj_group));
        *ngroups = (uint32_t)grp.size();
    }
    return CRT_OK;
}

extern "C" int crt_load_obj(const char *path, float **xyz_nrm, uint32_t *nverts, uint32_t **indices, uint32_t *ntris)
{
    return crt_load_obj_ex(path, xyz_nrm, nverts, indices, ntris, nullptr, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------------------------------------
static int get_env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

namespace crt_detail {
// Tuning knobs, read from the environment when a scene is committed (and by crt_scene_reload_tuning), not per frame.
struct Tuning {
    int kernel = 0;            // CRT_KERNEL: 0 render_kernel (pixel per lane), 1 render_pool_kernel (whole-pixel frames only)
    int pool_refill = 16;      // CRT_POOL_REFILL: pool kernel, free lanes that end a run of walk steps
    int sched_threshold = 0;   // CRT_SCHED_THRESHOLD: lanes that must wait for the shade phase before it runs; 0 = by scene: 44, or 52 for scenes
                               // whose tree holds sphere slots (scene.h:53-90: 10.65 against 10.85 ms; triangle scenes, rings and the grid: 44 is the optimum)
    int fetch_block = 0;       // CRT_FETCH_BLOCK: work units per trip to a block counter (a power of two, 64..65536); 0: chosen per frame
    int node_cache = 1;        // CRT_NODE_CACHE: 0 single waves, 1 chosen per frame, 2 four-wave workgroups with an LDS copy of the tree's top, 3 whole tree in one 16-wave workgroup, 4 16 + 4 wave split (crt_kernels.h)
    int stripe_bands = 1;      // CRT_STRIPE_BANDS: each block counter owns a contiguous band of the frame
    int waves_per_cu = 20;     // CRT_WAVES_PER_CU: persistent waves launched per compute unit (96 VGPRs: 5 per SIMD)
    int print_sections = 0;    // CRT_PRINT_SECTIONS: with CRT_COUNT_WORK and stats, print the section clocks
    int chunk_mode = 0;        // CRT_CHUNK_MODE: sample chunks through slabs + a second kernel (1), through the waves' rings (2), or chosen per frame (0)
    int slab_budget_mb = 2048; // CRT_SLAB_BUDGET_MB: the largest chunk scratch a frame may ask for before it takes the ring instead
    int ring_batch = 0;        // CRT_RING_BATCH: sample chunks, complete blocks a wave lets collect before one pass adds them up (0 = default)
    int ring_slots = 0;        // CRT_RING_SLOTS: sample chunks, blocks a wave may have in flight in its ring (2..16; 0 = default, crt_kernels.h)
    int bvh_wide = 0;          // CRT_BVH_WIDE: 1 = four-child nodes (crt_bvh.h: widen_bvh); read at commit only.  Off by default: measured slower
    void from_env()
    {
        kernel = get_env_int("CRT_KERNEL", 0) == 1 ? 1 : 0;
        pool_refill = std::min(std::max(get_env_int("CRT_POOL_REFILL", 16), 1), 63);
        sched_threshold = std::min(std::max(get_env_int("CRT_SCHED_THRESHOLD", 0), 0), 64);
        node_cache = std::min(std::max(get_env_int("CRT_NODE_CACHE", 1), 0), 4);
        fetch_block = std::min(std::max(get_env_int("CRT_FETCH_BLOCK", 0), 0), 65536);
        stripe_bands = get_env_int("CRT_STRIPE_BANDS", 1) ? 1 : 0;
        waves_per_cu = std::min(std::max(get_env_int("CRT_WAVES_PER_CU", 20), 1), 64);
        print_sections = get_env_int("CRT_PRINT_SECTIONS", 0);
        ring_slots = std::min(std::max(get_env_int("CRT_RING_SLOTS", 0), 0), 16);
        ring_batch = std::min(std::max(get_env_int("CRT_RING_BATCH", 0), 0), 8);
        chunk_mode = std::min(std::max(get_env_int("CRT_CHUNK_MODE", 0), 0), 2);
        slab_budget_mb = std::max(get_env_int("CRT_SLAB_BUDGET_MB", 2048), 0);
        bvh_wide = get_env_int("CRT_BVH_WIDE", 0) == 1 ? 1 : 0;
    }
};

// Per-stream scratch of one device: work counters, sample-chunk slabs (grow only: chunks x frame floats, summed in chunk
// order by finalize_chunks_kernel), the staging buffers of host-pointer renders (grow only) and the timing events.  Calls
// on one stream are ordered and share a set; calls on different streams get different sets, so their kernels may overlap
// (the next frame fills the GPU while this one's last waves drain) and two host-pointer renders never share staging.
// How a frame with sample chunks is rendered: through slabs and a second kernel, or through the waves' rings (crt_kernels.h: CM_*).
struct ChunkPlan { uint32_t mode = CM_WHOLE, ring_slots = 0, fetch_units = 0, ring_batch = 0; size_t bytes = 0; }; // (fetch_units / ring_batch 0: the tuning's, or the launcher's defaults)

struct StreamScratch {
    hipStream_t stream = nullptr; bool used = false, recorded = false; uint64_t last_use = 0;
    unsigned long long *d_counters = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr; // around the render kernel; after the last use of the set
    SplitStreams split{ nullptr, nullptr, nullptr };           // second stream + fork / join events for the two-launch frame (crt_kernels.hip: launch_split)
    float *d_chunks = nullptr; size_t chunk_bytes = 0; // sample chunks: the slabs [pixel][chunk][3] or the waves' rings (crt_kernels.h), whichever the frame uses
    ChunkPlan plan;                                      // ... of the frame last enqueued with this set
    float *d_stage_rgb = nullptr; size_t stage_rgb_floats = 0;
    int *d_stage_ids = nullptr; size_t stage_ids = 0;
};

// One copy of the committed scene on one HIP device.
struct DeviceScene {
    int device = -1;
    void *d_nodes = nullptr, *d_tris = nullptr, *d_spheres = nullptr, *d_shades = nullptr, *d_normals = nullptr;
    uint32_t cus = 0; // compute units: the persistent grid is cus x Tuning::waves_per_cu waves
    SceneView view{};
    // The helper stream of the two-launch frames (crt_kernels.hip: launch_split), ONE per device and scene, shared by every caller
    // stream, plus crt_render_noalloc's fork / join events (made at commit: that call may create nothing itself).  One helper for
    // all streams on purpose: with two frames in flight the small workgroups of frame k + 1 must not start before those of frame k
    // have left -- a CU that the large workgroup of frame k has just left would otherwise take several of them, and the large
    // workgroup of frame k + 1 would find no room there for the whole frame (measured: 8.8-9.2 ms per step instead of 8.3).
    SplitStreams noalloc_split{ nullptr, nullptr, nullptr };
    static constexpr int MAX_STREAMS = 4;
    StreamScratch scratch[MAX_STREAMS];
    uint64_t use_clock = 0;
    // The set of `stream`; a stream seen for the first time takes a free set or else the least recently used one, whose
    // last render it then waits for (*handover = that render's end event).
    StreamScratch *scratch_for(hipStream_t stream, hipEvent_t *handover)
    {
        *handover = nullptr;
        StreamScratch *pick = nullptr;
        for (StreamScratch &x : scratch) if (x.used && x.stream == stream) pick = &x;
        if (!pick) for (StreamScratch &x : scratch) if (!x.used) { pick = &x; break; }
        if (!pick) {
            pick = &scratch[0];
            for (StreamScratch &x : scratch) if (x.last_use < pick->last_use) pick = &x;
            if (pick->recorded) *handover = pick->ev_done;
        }
        pick->used = true; pick->stream = stream; pick->last_use = ++use_clock;
        return pick;
    }
    void release()
    {
        if (device >= 0) (void)hipSetDevice(device);
        void *ptrs[] = { d_nodes, d_tris, d_spheres, d_shades, d_normals };
        for (void *p : ptrs) if (p) (void)hipFree(p);
        d_nodes = d_tris = d_spheres = d_shades = d_normals = nullptr;
        if (noalloc_split.fork) (void)hipEventDestroy(noalloc_split.fork);
        if (noalloc_split.join) (void)hipEventDestroy(noalloc_split.join);
        if (noalloc_split.helper) (void)hipStreamDestroy(noalloc_split.helper);
        noalloc_split = SplitStreams{ nullptr, nullptr, nullptr }; // (the per-stream sets below only borrowed the helper stream)
        for (StreamScratch &x : scratch) {
            if (x.d_counters) (void)hipFree(x.d_counters);
            if (x.d_chunks) (void)hipFree(x.d_chunks);
            if (x.d_stage_rgb) (void)hipFree(x.d_stage_rgb);
            if (x.d_stage_ids) (void)hipFree(x.d_stage_ids);
            if (x.ev0) (void)hipEventDestroy(x.ev0);
            if (x.ev1) (void)hipEventDestroy(x.ev1);
            if (x.ev_done) (void)hipEventDestroy(x.ev_done);
            if (x.split.fork) (void)hipEventDestroy(x.split.fork);
            if (x.split.join) (void)hipEventDestroy(x.split.join);
            x = StreamScratch();
        }
        device = -1;
    }
};

// crt_render_multi's per-rank state: two sets of {render stream, local part, events} that consecutive frames alternate
// between, and on the root the gathered parts and the end-of-frame events.
struct MultiRank {
    hipStream_t stream[2] = { nullptr, nullptr };
    float *d_local[2] = { nullptr, nullptr }; size_t local_floats = 0; // compact tile list (tile mode) or raw sums (sample mode)
    float *d_all[2] = { nullptr, nullptr }; size_t all_floats = 0;      // the n parts back to back: on the root, and on every rank for ncclAllGather
    hipEvent_t sent[2] = { nullptr, nullptr };                          // copy transport: this rank's part has reached the root
    bool sent_recorded[2] = { false, false };
    void *comm = nullptr;                                               // ncclComm_t (RCCL transport)
};
// crt_render_multi's rank workers: one thread per logical rank beyond the first, started by crt_scene_commit_devices.  A frame's
// per-rank part of the enqueue (a dozen HIP calls: waits, counters, one or two launches with their fork / join, events) is issued
// by the rank's own thread, rank 0's by the caller's, so the host's time per frame is one rank's, not n ranks' (measured with 8
// logical ranks).  The exchange and the epilogue stay on the caller's thread.
struct RankWorkers {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable wake, done;
    uint64_t generation = 0;
    uint32_t pending = 0;
    bool stop = false;
    const std::function<int(uint32_t)> *job = nullptr;
    std::vector<int> rc;
    std::vector<std::string> msg;
    explicit RankWorkers(uint32_t n) : rc(n, 0), msg(n)
    {
        for (uint32_t r = 1; r < n; r++) threads.emplace_back([this, r] { run(r); });
    }
    ~RankWorkers()
    {
        { std::lock_guard<std::mutex> l(mu); stop = true; }
        wake.notify_all();
        for (std::thread &t : threads) t.join();
    }
    void run(uint32_t r);
    int run_all(const std::function<int(uint32_t)> &j); // j(r) for every rank; the first failure, with its message
};
struct MultiState {
    bool ready = false, use_rccl = false, peer_ids = false;
    std::unique_ptr<RankWorkers> workers;                               // null: the caller's thread issues every rank (CRT_MULTI_THREADS=0, or one rank)
    std::vector<MultiRank> rank;
    float *d_stage_rgb = nullptr; size_t stage_rgb_floats = 0;          // root: staging of host-pointer frames
    int *d_stage_ids = nullptr; size_t stage_ids = 0;
    hipEvent_t frame_done[2] = { nullptr, nullptr }; bool done_recorded[2] = { false, false };
    hipEvent_t fork = nullptr;                                          // the caller's stream at the start of a frame
    uint64_t frames = 0;
};
} // namespace crt_detail
using namespace crt_detail;

struct crt_scene {
    std::vector<HostTriangle> tris;
    std::vector<HostSphere> spheres;
    std::vector<Material> mats;
    std::vector<Texture> texs;
    // prepared (host: ranks + BVH) / committed (device buffers)
    bool prepared = false;
    bool committed = false;
    int device = -1; // = dev[0].device
    std::vector<uint32_t> rank;
    FlatBvh bvh;
    std::vector<Shade> shades; // materials joined with their textures (device record)
    std::vector<q4> normals;   // per triangle slot: unit normal + primitive id (crt_traverse.h: triangle_normal)
    uint32_t builder = 0, binary_depth = 0; float build_ms = 0.0f, sah_cost = 0.0f;
    // A tree built on the device leaves its slots and normals THERE (no host copy: bvh.tris / normals stay empty, nslots counts):
    // the replica on that device adopts the two buffers, further replicas copy from that one device to device.
    struct Built { int device = -1; void *d_tris = nullptr, *d_normals = nullptr; } built;
    size_t nslots = 0;
    std::vector<Sphere> flat_spheres; // the spheres that are tested before the walk (the others are slots of the tree)
    bool sphere_slots = false;        // some spheres went into the tree
    float sphere_slot_rmax = 0.0f;    // the largest radius among them (SceneView::sphere_slots carries it)
    Tuning tuning;
    std::vector<DeviceScene> dev;     // dev[0]: the copy crt_render uses; one per logical rank after crt_scene_commit_devices
    MultiState multi;
    std::mutex mutex;                 // crt_render's bookkeeping (scratch sets, staging): calls on different streams may come from different threads

    void release_multi();
    void release_built()
    {
        if (built.d_tris || built.d_normals) {
            if (built.device >= 0) (void)hipSetDevice(built.device);
            if (built.d_tris) (void)hipFree(built.d_tris);
            if (built.d_normals) (void)hipFree(built.d_normals);
        }
        built = Built();
    }
    void release_device()
    {
        release_multi();
        release_built();
        for (DeviceScene &d : dev) d.release();
        dev.clear();
        device = -1;
        committed = false;
    }
};

extern "C" crt_scene *crt_scene_create(void) { return new (std::nothrow) crt_scene(); }
extern "C" void crt_scene_destroy(crt_scene *s)
{
    if (!s) return;
    s->release_device();
    delete s;
}

extern "C" int crt_scene_add_texture(crt_scene *s, int type, const float a[3], const float b[3], float scale, uint32_t *out_id)
{
    if (!s || !a) return fail(CRT_ERR_INVALID, "crt_scene_add_texture: null argument");
    if (type != CRT_TEX_SOLID && type != CRT_TEX_CHECKER) return fail(CRT_ERR_INVALID, "crt_scene_add_texture: unknown type");
    if (type == CRT_TEX_CHECKER && !b) return fail(CRT_ERR_INVALID, "crt_scene_add_texture: checker needs two colours");
    Texture t{};
    t.type = type == CRT_TEX_SOLID ? TEX_SOLID : TEX_CHECKER;
    t.ar = a[0]; t.ag = a[1]; t.ab = a[2];
    if (b) { t.br = b[0]; t.bg = b[1]; t.bb = b[2]; }
    t.inv_scale = type == CRT_TEX_CHECKER ? 1.f / scale : 0.0f; // texture.h:26
    s->texs.push_back(t);
    if (out_id) *out_id = (uint32_t)s->texs.size() - 1;
    s->committed = false; s->prepared = false;
    return CRT_OK;
}

extern "C" int crt_scene_add_material(crt_scene *s, int type, const float rgb[3], float param, uint32_t texture_id, uint32_t *out_id)
{
    if (!s || !rgb) return fail(CRT_ERR_INVALID, "crt_scene_add_material: null argument");
    if (type < CRT_MAT_LAMBERTIAN || type > CRT_MAT_DIELECTRIC) return fail(CRT_ERR_INVALID, "crt_scene_add_material: unknown type");
    const bool textured = type == CRT_MAT_LAMBERTIAN || type == CRT_MAT_DIFFUSE_LIGHT;
    if (textured && texture_id == UINT32_MAX) { // material.h:22,107: new SolidColor(color)
        int rc = crt_scene_add_texture(s, CRT_TEX_SOLID, rgb, nullptr, 1.0f, &texture_id);
        if (rc) return rc;
    }
    if (textured && texture_id >= s->texs.size()) return fail(CRT_ERR_INVALID, "crt_scene_add_material: texture id out of range");
    Material m{};
    m.type = type; m.r = rgb[0]; m.g = rgb[1]; m.b = rgb[2]; m.param = param; m.tex = textured ? (int)texture_id : 0;
    s->mats.push_back(m);
    if (out_id) *out_id = (uint32_t)s->mats.size() - 1;
    s->committed = false; s->prepared = false;
    return CRT_OK;
}

extern "C" int crt_scene_add_mesh(crt_scene *s, const float *xyz, uint32_t stride_floats, uint32_t nverts, const uint32_t *indices, uint32_t ntris,
                                  uint32_t material_id)
{
    if (!s || !xyz || !indices) return fail(CRT_ERR_INVALID, "crt_scene_add_mesh: null argument");
    if (stride_floats < 3) return fail(CRT_ERR_INVALID, "crt_scene_add_mesh: stride_floats < 3");
    if (material_id >= s->mats.size()) return fail(CRT_ERR_INVALID, "crt_scene_add_mesh: material id out of range");
    if ((uint64_t)s->tris.size() + ntris > (1u << 28)) return fail(CRT_ERR_LIMIT, "crt_scene_add_mesh: more than 2^28 triangles");
    for (uint32_t t = 0; t < ntris; t++) // scene.h:134-136 bounds check, here an error instead of a silent skip
        for (int k = 0; k < 3; k++)
            if (indices[3 * (size_t)t + k] >= nverts) return fail(CRT_ERR_INVALID, "crt_scene_add_mesh: vertex index out of range");
    s->tris.reserve(s->tris.size() + ntris);
    for (uint32_t t = 0; t < ntris; t++) { // scene.h:128-144
        const float *p0 = xyz + (size_t)indices[3 * (size_t)t] * stride_floats;
        const float *p1 = xyz + (size_t)indices[3 * (size_t)t + 1] * stride_floats;
        const float *p2 = xyz + (size_t)indices[3 * (size_t)t + 2] * stride_floats;
        HostTriangle tri;
        tri.a = mk3(p0[0], p0[1], p0[2]); tri.b = mk3(p1[0], p1[1], p1[2]); tri.c = mk3(p2[0], p2[1], p2[2]);
        tri.material = (int)material_id;
        s->tris.push_back(tri);
    }
    s->committed = false; s->prepared = false;
    return CRT_OK;
}

extern "C" int crt_scene_add_sphere(crt_scene *s, const float center[3], float radius, uint32_t material_id)
{
    if (!s || !center) return fail(CRT_ERR_INVALID, "crt_scene_add_sphere: null argument");
    if (material_id >= s->mats.size()) return fail(CRT_ERR_INVALID, "crt_scene_add_sphere: material id out of range");
    HostSphere sp;
    sp.center = mk3(center[0], center[1], center[2]);
    sp.radius_arg = radius;
    sp.radius = sphere_radius(radius); // sphere.h:11 std::max(0.0f, radius)
    sp.material = (int)material_id;
    s->spheres.push_back(sp);
    s->committed = false; s->prepared = false;
    return CRT_OK;
}

// CRT_PRINT_COMMIT=1: where a commit's wall time goes, stage by stage, on stderr
struct CommitClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    const bool on = getenv("CRT_PRINT_COMMIT") != nullptr;
    void lap(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "crt commit: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

template <class T>
static int upload(void **dst, const std::vector<T> &v)
{
    const size_t bytes = std::max<size_t>(v.size() * sizeof(T), 64);
    HIP_TRY(hipMalloc(dst, bytes));
    if (!v.empty()) HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return CRT_OK;
}

// GPU builder (crt_lbvh.hip): ranks + LBVH on the device.  Returns false (with everything untouched) when the scene or the
// resulting tree does not qualify, so that the caller falls back to the host path.
static bool commit_gpu_lbvh(crt_scene *s, int device, uint32_t max_leaf, uint32_t tree_kind)
{
    if (device < 0 || s->tris.empty() || s->spheres.size() > 8 || s->tris.size() >= (1u << 28)) return false;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev || hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return false; }
    CommitClock clk;
    const size_t n = s->tris.size();
    std::vector<float> tri9(n * 9);
    std::vector<int> mat(n);
    for (size_t i = 0; i < n; i++) {
        const HostTriangle &t = s->tris[i];
        float *v = &tri9[9 * i];
        v[0] = t.a.x; v[1] = t.a.y; v[2] = t.a.z; v[3] = t.b.x; v[4] = t.b.y; v[5] = t.b.z; v[6] = t.c.x; v[7] = t.c.y; v[8] = t.c.z;
        mat[i] = t.material;
    }
    std::vector<LbvhSphereInput> sph(s->spheres.size());
    Box su{};
    for (size_t k = 0; k < s->spheres.size(); k++) {
        sph[k].refbox = reference_sphere_box(s->spheres[k]);
        if (k == 0) su = sph[k].refbox;
        else { // interval.h:11-14 per axis
            const Box &b = sph[k].refbox;
            su.mn.x = su.mn.x <= b.mn.x ? su.mn.x : b.mn.x; su.mx.x = su.mx.x >= b.mx.x ? su.mx.x : b.mx.x;
            su.mn.y = su.mn.y <= b.mn.y ? su.mn.y : b.mn.y; su.mx.y = su.mx.y >= b.mx.y ? su.mx.y : b.mx.y;
            su.mn.z = su.mn.z <= b.mn.z ? su.mn.z : b.mn.z; su.mx.z = su.mx.z >= b.mx.z ? su.mx.z : b.mx.z;
        }
    }
    LbvhOutput out;
    out.keep_on_device = tree_kind == LBVH_TREE_SAH && !getenv("CRT_BUILD_HOST_COPIES"); // (the debugging switch brings everything to the host as before)
    clk.lap("pack triangles");
    if (lbvh_build(tri9.data(), mat.data(), (uint32_t)n, su, !sph.empty(), sph.data(), (uint32_t)sph.size(), max_leaf, tree_kind, out, nullptr) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    clk.lap("device build + transfers");
    if (out.depth > CRT_STACK_DEPTH) { // long runs of equal codes: the host builder bounds the depth
        if (out.d_slots) (void)hipFree(out.d_slots);
        if (out.d_normals) (void)hipFree(out.d_normals);
        return false;
    }
    // ranks: triangles from the device; a sphere follows every triangle with a code <= its own (stable sort, triangles
    // first in the reference's list) and the spheres that sort before it
    s->rank.assign(n + s->spheres.size(), 0);
    std::copy(out.rank_of_tri.begin(), out.rank_of_tri.end(), s->rank.begin());
    for (size_t k = 0; k < s->spheres.size(); k++) {
        const uint32_t ck = out.sphere_codes.code[k];
        uint32_t r = (uint32_t)(std::upper_bound(out.sorted_codes.begin(), out.sorted_codes.end(), ck) - out.sorted_codes.begin());
        for (size_t j = 0; j < s->spheres.size(); j++) r += (out.sphere_codes.code[j] < ck || (out.sphere_codes.code[j] == ck && j < k)) ? 1u : 0u;
        s->rank[n + k] = r;
    }
    s->bvh = FlatBvh();
    if (!out.breadth_first) renumber_breadth_first(out.nodes, out.root_ref); // Karras numbering -> breadth first (the top of the tree goes to LDS)
    s->normals.swap(out.normals); // (SAH builder: made on the device with the slots; else empty)
    s->bvh.nodes.swap(out.nodes);
    s->bvh.tris.swap(out.slots);
    s->nslots = n;
    s->built.device = device; s->built.d_tris = out.d_slots; s->built.d_normals = out.d_normals;
    s->bvh.root_ref = out.root_ref; s->bvh.root_box = out.root_box; s->bvh.depth = out.depth; s->bvh.max_leaf = max_leaf;
    s->builder = tree_kind == LBVH_TREE_SAH ? 2u : 1u; s->build_ms = out.build_ms;
    clk.lap("sphere ranks + renumbering");
    return true;
}

extern "C" int crt_scene_commit(crt_scene *s, int device)
{
    const char *b = getenv("CRT_BUILDER");
    uint32_t flags = CRT_BUILD_AUTO;
    if (b && strcmp(b, "lbvh") == 0) flags = CRT_BUILD_GPU_LBVH;
    else if (b && strcmp(b, "gpusah") == 0) flags = CRT_BUILD_GPU_SAH;
    else if (b && strcmp(b, "host") == 0) flags = 0u;
    return crt_scene_commit_ex(s, device, flags);
}

// CRT_BUILD_AUTO: the device builds the SAH tree of a scene large enough for the host to take longer (crt_gsah.hip: the same
// tree statistics in 10 ms instead of 630 ms for a million triangles; below a few ten thousand triangles the host is as fast)
static uint32_t resolve_builder(const crt_scene *s, int device, uint32_t flags)
{
    if (!(flags & CRT_BUILD_AUTO)) return flags;
    flags &= ~(uint32_t)CRT_BUILD_AUTO;
    if (flags & (CRT_BUILD_GPU_LBVH | CRT_BUILD_GPU_SAH)) return flags;
    return (device >= 0 && s->tris.size() >= 32768 && s->spheres.size() <= 8) ? flags | CRT_BUILD_GPU_SAH : flags;
}

// Host side of a commit: ranks, tree, shading records, normals (with CRT_BUILD_GPU_LBVH ranks and tree come from
// `build_device`).  Leaves s->prepared set.
static int prepare_host(crt_scene *s, int build_device, uint32_t flags)
{
    if (s->tris.empty() && s->spheres.empty()) return fail(CRT_ERR_INVALID, "crt_scene_commit: empty scene");
    s->release_device();
    s->prepared = false;
    s->tuning.from_env();

    const uint32_t max_leaf = (uint32_t)get_env_int("CRT_MAX_LEAF", 2);
    s->builder = 0; s->build_ms = 0.0f;
    flags = resolve_builder(s, build_device, flags);
    CommitClock clk;
    const bool gpu_built = (flags & (CRT_BUILD_GPU_LBVH | CRT_BUILD_GPU_SAH)) &&
                           commit_gpu_lbvh(s, build_device, std::min(std::max(max_leaf, 1u), 8u), (flags & CRT_BUILD_GPU_SAH) ? LBVH_TREE_SAH : LBVH_TREE_KARRAS);
    const auto host_t0 = std::chrono::steady_clock::now();
    if (!gpu_built) {
        // reference order: all triangles, then all spheres (scene.h:161-165)
        std::vector<Box> boxes;
        boxes.reserve(s->tris.size() + s->spheres.size());
        for (const HostTriangle &t : s->tris) boxes.push_back(reference_triangle_box(t));
        for (const HostSphere &sp : s->spheres) boxes.push_back(reference_sphere_box(sp));
        Box scene_box;
        reference_ranks(boxes, s->rank, scene_box);
    }

    // Where the spheres go.  A few spheres (the reference's triangle scenes have two, scene.h:164-165) stay a flat list
    // tested before the walk, so the r = 1000 ground never inflates the tree (F10).  A scene made of spheres
    // (scene.h:53-90: 488 of them) puts them into the tree as sphere slots, except those far larger than the typical one.
    std::vector<BvhSphere> tree_spheres;
    std::vector<Sphere> &flat = s->flat_spheres;
    flat.clear();
    float big = FLT_MAX;
    if (s->spheres.size() > 8) {
        std::vector<float> radii;
        for (const HostSphere &sp : s->spheres) radii.push_back(sp.radius);
        std::nth_element(radii.begin(), radii.begin() + (long)(radii.size() / 2), radii.end());
        big = 16.0f * radii[radii.size() / 2];
    }
    for (HostSphere &sp : s->spheres) sp.inv_radius = reciprocal_if_exact(sp.radius); // for the hit normal's (p - c) / radius (sphere.h:45)
    for (size_t k = 0; k < s->spheres.size(); k++) {
        const HostSphere &sp = s->spheres[k];
        const uint32_t id = (uint32_t)(s->tris.size() + k), rank = s->rank[s->tris.size() + k];
        if (s->spheres.size() > 8 && sp.radius <= big) {
            tree_spheres.push_back(BvhSphere{ sp, id, rank });
        } else {
            Sphere o;
            o.cx = sp.center.x; o.cy = sp.center.y; o.cz = sp.center.z; o.radius_arg = sp.radius_arg;
            o.id = (int)id; o.rank = (int)rank; o.mat = sp.material; o.inv_radius = sp.inv_radius; o.sure = sphere_sure_term(sp.center, sp.radius); o.radius = sp.radius; o.pad1 = o.pad2 = 0;
            flat.push_back(o);
        }
    }
    if (flat.size() > 4096) return fail(CRT_ERR_LIMIT, "crt_scene_commit: more than 4096 spheres outside the tree");
    s->sphere_slots = !tree_spheres.empty();
    s->sphere_slot_rmax = FLT_MIN;
    for (const BvhSphere &sp : tree_spheres) s->sphere_slot_rmax = std::max(s->sphere_slot_rmax, sp.s.radius);

    if (!gpu_built) {
        build_bvh(s->tris, s->rank, tree_spheres, max_leaf, s->bvh);
        s->build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    }
    clk.lap("ranks + tree");
    if (s->bvh.depth > CRT_STACK_DEPTH) return fail(CRT_ERR_LIMIT, "crt_scene_commit: BVH deeper than the traversal stack");
    // Four children per node (CRT_BVH_WIDE=1): half the levels, half the node fetches per ray, and 2.3 times the instructions per
    // fetch -- the kernel is bound by instruction issue, not by the latency of dependent fetches, so the two-child format stays
    // the default (DESIGN.md section 4 has the measurement).  The kernels for this format have the 16-level stack
    // (crt_kernels.hip: launch_wide); a tree that stays deeper than that keeps two children.
    s->binary_depth = s->bvh.depth;
    s->sah_cost = bvh_sah_cost(s->bvh);
    if (s->tuning.bvh_wide == 1) {
        FlatBvh wide = s->bvh;
        widen_bvh(wide);
        if (wide.wide && wide.depth <= 16) s->bvh = std::move(wide);
    }

    s->shades.assign(s->mats.size(), Shade{});
    for (size_t k = 0; k < s->mats.size(); k++) {
        const Material &m = s->mats[k];
        Shade &o = s->shades[k];
        o.type = m.type; o.param = m.param; o.tex_type = TEX_SOLID;
        if (m.type == MAT_LAMBERTIAN || m.type == MAT_DIFFUSE_LIGHT) { // the materials that own a texture (material.h:22,107)
            const Texture &t = s->texs[(size_t)m.tex];
            o.tex_type = t.type; o.inv_scale = t.inv_scale;
            o.ar = t.ar; o.ag = t.ag; o.ab = t.ab; o.br = t.br; o.bg = t.bg; o.bb = t.bb;
        } else if (m.type == MAT_METAL) { // material.h:48
            o.ar = m.r; o.ag = m.g; o.ab = m.b;
        } else { // Dielectric, material.h:66
            o.ar = o.ag = o.ab = 1.0f;
        }
    }

    // the unit normal of every triangle slot, with the operations the hit record uses (sphere slots: unused)
    {
        if (!gpu_built) s->nslots = s->bvh.tris.size() / 3;
        const size_t nslots = s->nslots;
        const bool have = gpu_built && (s->built.d_normals || (s->normals.size() == nslots && nslots > 0));
        if (!have) s->normals.resize(nslots);
        for (size_t k = 0; k < nslots && !have; k++) {
            const q4 *tp = &s->bvh.tris[3 * k];
            int rank_word;
            std::memcpy(&rank_word, &tp[1].w, 4);
            if (rank_word & CRT_SPHERE_SLOT) { q4 z; z.x = z.y = z.z = 0.0f; z.w = tp[0].w; s->normals[k] = z; }
            else s->normals[k] = triangle_normal(tp[0], tp[1], tp[2]);
        }
    }
    clk.lap("shading records + normals");
    s->prepared = true;
    return CRT_OK;
}

// The fork / join events of a two-launch frame (crt_kernels.hip: launch_split) around the device's helper stream (nullptr: made here).
static hipError_t make_split_streams(SplitStreams &sp, hipStream_t helper)
{
    hipError_t e;
    sp.helper = helper;
    if (!sp.helper && (e = hipStreamCreateWithFlags(&sp.helper, hipStreamNonBlocking)) != hipSuccess) return e;
    if ((e = hipEventCreateWithFlags(&sp.fork, hipEventDisableTiming)) != hipSuccess) return e;
    return hipEventCreateWithFlags(&sp.join, hipEventDisableTiming);
}

// Device side of a commit: one copy of the prepared scene on `device`.
static int upload_replica(crt_scene *s, DeviceScene &d, int device)
{
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CRT_ERR_NO_DEVICE, "crt_scene_commit: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    d.device = device;
    int rc;
    if ((rc = upload(&d.d_nodes, s->bvh.nodes))) return rc;
    if ((rc = upload(&d.d_spheres, s->flat_spheres))) return rc;
    if ((rc = upload(&d.d_shades, s->shades))) return rc;
    if (s->built.d_tris && s->built.device == device) { // the builder's own buffers, where it left them
        d.d_tris = s->built.d_tris; d.d_normals = s->built.d_normals;
        s->built = crt_scene::Built();
    } else if (s->bvh.tris.empty() && s->nslots && !s->dev.empty() && s->dev[0].d_tris && &d != &s->dev[0]) { // no host copy: from the first replica
        const size_t tb = s->nslots * 3 * sizeof(q4), nb = s->nslots * sizeof(q4);
        HIP_TRY(hipMalloc(&d.d_tris, tb));
        HIP_TRY(hipMalloc(&d.d_normals, nb));
        if (s->dev[0].device == device) {
            HIP_TRY(hipMemcpy(d.d_tris, s->dev[0].d_tris, tb, hipMemcpyDeviceToDevice));
            HIP_TRY(hipMemcpy(d.d_normals, s->dev[0].d_normals, nb, hipMemcpyDeviceToDevice));
        } else {
            HIP_TRY(hipMemcpyPeer(d.d_tris, device, s->dev[0].d_tris, s->dev[0].device, tb));
            HIP_TRY(hipMemcpyPeer(d.d_normals, device, s->dev[0].d_normals, s->dev[0].device, nb));
        }
    } else {
        if ((rc = upload(&d.d_tris, s->bvh.tris))) return rc;
        if ((rc = upload(&d.d_normals, s->normals))) return rc;
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    d.cus = (uint32_t)prop.multiProcessorCount;
    HIP_TRY(noalloc_scratch_init(device)); // the device's copy of the static scratch of crt_render_noalloc (crt_kernels.hip)
    (void)prepare_split_kernels();         // (a refusal only costs the two-launch form: launch_split then falls back)
    if (!d.noalloc_split.helper) {         // ... and the second stream of its two-launch frames (crt_kernels.hip: launch_split)
        HIP_TRY(make_split_streams(d.noalloc_split, nullptr));
    }
    SceneView &v = d.view;
    v.nodes = (const q4 *)d.d_nodes; v.tris = (const q4 *)d.d_tris; v.spheres = (const Sphere *)d.d_spheres;
    v.shades = (const Shade *)d.d_shades; v.normals = (const q4 *)d.d_normals;
    v.root_ref = s->bvh.root_ref; v.wide = s->bvh.wide ? 1u : 0u; v.nnodes = (uint32_t)(s->bvh.nodes.size() / (s->bvh.wide ? 8 : 4)); v.ntris = (uint32_t)s->nslots;
    v.nspheres = (uint32_t)s->flat_spheres.size(); v.sphere_slots = 0u;
    if (s->sphere_slots) std::memcpy(&v.sphere_slots, &s->sphere_slot_rmax, 4);
    v.root_min = s->bvh.root_box.mn; v.root_max = s->bvh.root_box.mx;
    return CRT_OK;
}

extern "C" int crt_scene_commit_ex(crt_scene *s, int device, uint32_t flags)
{
    if (!s) return fail(CRT_ERR_INVALID, "crt_scene_commit: null scene");
    int rc = prepare_host(s, device, flags);
    if (rc) return rc;
    if (device == CRT_DEVICE_NONE) return CRT_OK; // host-side preparation only (ranks, BVH); rendering stays unavailable
    s->dev.resize(1);
    CommitClock clk;
    if ((rc = upload_replica(s, s->dev[0], device))) { s->release_device(); return rc; }
    clk.lap("upload");
    s->device = device;
    s->committed = true;
    return CRT_OK;
}

extern "C" int crt_scene_reload_tuning(crt_scene *s)
{
    if (!s) return fail(CRT_ERR_INVALID, "crt_scene_reload_tuning: null scene");
    s->tuning.from_env();
    return CRT_OK;
}

extern "C" int crt_scene_get_info(const crt_scene *s, crt_scene_info *info)
{
    if (!s || !info) return fail(CRT_ERR_INVALID, "crt_scene_get_info: null argument");
    if (!s->prepared) return fail(CRT_ERR_INVALID, "crt_scene_get_info: scene not committed");
    memset(info, 0, sizeof *info);
    info->num_triangles = (uint32_t)s->tris.size(); info->num_spheres = (uint32_t)s->spheres.size();
    info->num_materials = (uint32_t)s->mats.size(); info->num_textures = (uint32_t)s->texs.size();
    info->num_nodes = (uint32_t)(s->bvh.nodes.size() / (s->bvh.wide ? 8 : 4)); info->bvh_depth = s->bvh.depth; info->max_leaf_size = s->bvh.max_leaf;
    info->device = s->committed ? (uint32_t)s->device : UINT32_MAX;
    info->node_bytes = s->bvh.nodes.size() * sizeof(q4); info->triangle_bytes = s->nslots * 3 * sizeof(q4);
    info->sphere_bytes = s->flat_spheres.size() * sizeof(Sphere);
    info->bounds_min[0] = s->bvh.root_box.mn.x; info->bounds_min[1] = s->bvh.root_box.mn.y; info->bounds_min[2] = s->bvh.root_box.mn.z;
    info->builder = s->builder; info->build_ms = s->build_ms; info->node_width = s->bvh.wide ? 4u : 2u; info->binary_depth = s->binary_depth; info->sah_cost = s->sah_cost;
    info->bounds_max[0] = s->bvh.root_box.mx.x; info->bounds_max[1] = s->bvh.root_box.mx.y; info->bounds_max[2] = s->bvh.root_box.mx.z;
    return CRT_OK;
}

extern "C" int crt_scene_get_ranks(const crt_scene *s, uint32_t *rank, uint32_t n)
{
    if (!s || !rank) return fail(CRT_ERR_INVALID, "crt_scene_get_ranks: null argument");
    if (!s->prepared) return fail(CRT_ERR_INVALID, "crt_scene_get_ranks: scene not committed");
    if (n != s->rank.size()) return fail(CRT_ERR_INVALID, "crt_scene_get_ranks: n != triangles + spheres");
    memcpy(rank, s->rank.data(), n * sizeof(uint32_t));
    return CRT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// camera (camera.h:25-58): evaluated on the host, handed to the kernel by value
// ---------------------------------------------------------------------------------------------------------
static inline float degrees_to_radians(float degrees) { return (float)((double)degrees * 3.14159265358979323846 / (double)180.0f); } // utility.h:67-69

extern "C" int crt_camera_init(crt_camera *cam, uint32_t width, const float lookfrom[3], const float lookat[3], const float vup_[3], float vfov,
                               float aspect, float defocus_angle, float focus_dist)
{
    if (!cam || !lookfrom || !lookat || !vup_) return fail(CRT_ERR_INVALID, "crt_camera_init: null argument");
    (void)aspect; // camera.h:27 uses the member default 16/9, not the argument (F12)
    const float aspect_ratio = 16.0f / 9.0f;
    uint32_t image_height = (uint32_t)((float)width / aspect_ratio);
    image_height = image_height < 1 ? 1 : image_height;
    const f3 look_from = mk3(lookfrom[0], lookfrom[1], lookfrom[2]), look_at = mk3(lookat[0], lookat[1], lookat[2]), vup = mk3(vup_[0], vup_[1], vup_[2]);
    const float theta = degrees_to_radians(vfov);
    const float h = std::tan(theta / 2.0f);
    const float viewport_height = 2.0f * h * focus_dist;
    const float viewport_width = viewport_height * ((float)width / (float)image_height);
    const f3 w = normalize(look_from - look_at);
    const f3 u = normalize(cross(vup, w));
    const f3 v = cross(w, u);
    const f3 viewport_horizontal = u * viewport_width;
    const f3 viewport_vertical = (-v) * viewport_height;
    const f3 pdu = div3(viewport_horizontal, (float)width);
    const f3 pdv = div3(viewport_vertical, (float)image_height);
    const f3 upper_left = ((look_from - w * focus_dist) - viewport_horizontal * 0.5f) - viewport_vertical * 0.5f;
    const f3 pixel00 = upper_left + (pdu + pdv) * 0.5f;
    const float defocus_radius = focus_dist * std::tan(degrees_to_radians(defocus_angle / 2.0f));
    const f3 disk_u = u * defocus_radius, disk_v = v * defocus_radius;
    cam->image_width = width; cam->image_height = image_height;
    cam->defocus_angle = 0.6f; // the constructor parameter shadows the member: get_ray tests the default (camera.h:20,94)
    cam->focus_distance = focus_dist;
    auto put = [](float dst[3], f3 s) { dst[0] = s.x; dst[1] = s.y; dst[2] = s.z; };
    put(cam->center, look_from); put(cam->pixel00, pixel00); put(cam->pixel_delta_u, pdu); put(cam->pixel_delta_v, pdv);
    put(cam->u, u); put(cam->v, v); put(cam->w, w); put(cam->defocus_disk_u, disk_u); put(cam->defocus_disk_v, disk_v);
    put(cam->background, mk3(0.0f, 0.0f, 0.0f)); // camera.h:23
    return CRT_OK;
}

static_assert(sizeof(crt_camera) == sizeof(CameraPOD), "crt_camera and CameraPOD must have the same layout");

// ---------------------------------------------------------------------------------------------------------
// render
// ---------------------------------------------------------------------------------------------------------
static bool is_device_pointer(const void *p)
{
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof attr);
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeUnified;
}

extern "C" uint64_t crt_compact_floats(uint32_t width, uint32_t height, uint32_t tile_ranks)
{
    if (tile_ranks == 0) return 0;
    const uint64_t ntiles = (uint64_t)((width + 7u) / 8u) * ((height + 7u) / 8u);
    return ((ntiles + tile_ranks - 1) / tile_ranks) * 64u * 3u;
}

// ---------------------------------------------------------------------------------------------------------
// what crt_render, crt_render_noalloc and crt_render_multi share
// ---------------------------------------------------------------------------------------------------------
static int check_render_params(const crt_scene *s, const crt_camera *cam, const crt_render_params *p, const void *out_rgb, const char *who, uint32_t *chunks_out)
{
    const std::string w(who);
    if (!s || !cam || !p || !out_rgb) return fail(CRT_ERR_INVALID, w + ": null argument");
    if (!s->committed) return fail(s->prepared ? CRT_ERR_NO_DEVICE : CRT_ERR_INVALID, w + ": scene not committed to a HIP device");
    if (p->width == 0 || p->height == 0 || p->spp == 0) return fail(CRT_ERR_INVALID, w + ": empty frame or zero samples");
    if (p->tile_ranks == 0 || p->tile_rank >= p->tile_ranks) return fail(CRT_ERR_INVALID, w + ": tile_rank must be < tile_ranks");
    if ((uint64_t)p->width * p->height > (1ull << 31)) return fail(CRT_ERR_LIMIT, w + ": more than 2^31 pixels");
    if (p->width > 65535u || p->height > 65535u || p->spp > 65535u || p->max_depth > 255u)
        return fail(CRT_ERR_LIMIT, w + ": width, height, spp must be <= 65535 and max_depth <= 255 (packed lane state)");
    const uint32_t chunks = p->sample_chunks > 1 ? p->sample_chunks : 1u;
    if (chunks > 1 && p->spp % chunks != 0) return fail(CRT_ERR_INVALID, w + ": sample_chunks must divide spp");
    *chunks_out = chunks;
    return CRT_OK;
}

static size_t output_floats(const crt_render_params *p)
{
    return (p->flags & CRT_OUT_COMPACT) ? (size_t)crt_compact_floats(p->width, p->height, p->tile_ranks) : (size_t)p->width * p->height * 3;
}

// The frame's kernel arguments as far as the caller decides them (launch_render completes the work queue's geometry).
static RenderArgs render_args(const crt_scene *s, const crt_render_params *p, uint32_t chunks)
{
    const Tuning &t = s->tuning;
    RenderArgs ra{};
    ra.width = p->width; ra.height = p->height; ra.tiles_x = (p->width + 7u) / 8u; ra.tiles_y = (p->height + 7u) / 8u;
    ra.spp = p->spp; ra.sample_offset = p->sample_offset; ra.max_depth = p->max_depth; ra.ids_sample = p->ids_sample;
    ra.tile_rank = p->tile_rank; ra.tile_ranks = p->tile_ranks;
    ra.flags = ((p->flags & CRT_OUT_ACCUM) ? (uint32_t)RA_ACCUM : 0u) | ((p->flags & CRT_OUT_COMPACT) ? (uint32_t)RA_COMPACT : 0u);
    ra.key0 = (uint32_t)p->seed; ra.key1 = (uint32_t)(p->seed >> 32);
    // the pool kernel has no sample chunks: chosen here with launch_render's own condition, and the threshold with it
    const bool pool = t.kernel == 1 && chunks == 1 && pool_kernel_built();
    ra.kernel = pool ? (uint32_t)RK_POOL : (uint32_t)RK_LANES;
    ra.sched_threshold = (uint32_t)(pool ? t.pool_refill : t.sched_threshold ? t.sched_threshold : s->sphere_slots ? 52 : 44);
    ra.bvh_depth = s->bvh.depth;
    ra.node_cache = (uint32_t)t.node_cache;
    ra.fetch_units = (uint32_t)t.fetch_block;         // 0 = launch_render's rule
    ra.blocks_per_stripe = t.stripe_bands ? 1u : 0u; // bands by default (+0.5 % cats, +2 % on the 1 M-triangle grid); launch_render sets the count
    ra.chunks = chunks;
    ra.chunk_mode = chunks > 1 ? (uint32_t)CM_RING : (uint32_t)CM_WHOLE; // (plan_chunks decides between ring and slabs)
    ra.ring_slots = (uint32_t)t.ring_slots;
    ra.ring_batch = (uint32_t)t.ring_batch;
    return ra;
}

// Sample chunks, two ways (same frame bit for bit).  Slabs: every chunk's raw sum goes to a scratch of pixels x chunks x 12 bytes and
// a second kernel adds a pixel's chunk sums -- the fastest (7.9 ms on the headline frame against 8.4 through the rings: the waves
// neither count nor add), but 1.6 GB for 1080p x 64 chunks and 25 GB for 1080p x 1024.  Rings: the wave that rendered a block of
// pixels adds them up itself through a ring of a few KB per wave, a fixed 30 - 130 MB per stream whatever the frame, any chunk count.
// Chosen per frame: slabs while they fit `budget` bytes (Tuning::slab_budget_mb; crt_render_noalloc: its static scratch) and the
// frame is whole or compact (the second kernel writes every pixel of `out`), else the ring with as many slots as fit.
static ChunkPlan plan_chunks(const crt_scene *s, const DeviceScene &d, const crt_render_params *p, uint32_t chunks, size_t budget, bool ring_must_fit)
{
    ChunkPlan plan;
    if (chunks <= 1) return plan;
    const Tuning &t = s->tuning;
    const size_t slab_bytes = output_floats(p) * (size_t)chunks * sizeof(float);
    const bool slab_ok = !(p->tile_ranks > 1 && !(p->flags & CRT_OUT_COMPACT)) && slab_bytes <= budget && (uint64_t)(output_floats(p) / 3) * chunks < (1ull << 32); // (32-bit unit index)
    if (t.chunk_mode == 1 ? slab_ok : (t.chunk_mode == 0 && slab_ok)) { plan.mode = CM_SLAB; plan.bytes = slab_bytes; return plan; }
    RenderArgs ra = render_args(s, p, chunks);
    ra.chunk_mode = CM_RING;
    const uint32_t waves = d.cus * (uint32_t)t.waves_per_cu;
    plan.mode = CM_RING;
    if (ring_must_fit) { // crt_render_noalloc: what its static scratch admits (blocks of 128 units, 2..4 slots, no batching)
        plan.ring_slots = render_ring_slots_that_fit(ra, waves, budget);
    } else {
        // crt_render, where memory is no object (126 MB per stream on 256 CUs at 64 spp): 8 slots of 256-unit blocks, complete blocks
        // added up four at a time -- 8.42 ms on the headline frame against 9.43 with 4 x 128 x 1 (16 slots: 8.38;
        // profiles/r3_v2_gpu_sah/ringsweep.txt).  An environment setting wins; very large sample counts (a block is at least
        // one pixel's samples) take fewer slots so that the ring stays within the slab budget where it can.
        if (t.fetch_block == 0) plan.fetch_units = ra.fetch_units = 256u;
        plan.ring_batch = t.ring_batch ? (uint32_t)t.ring_batch : 4u;
        plan.ring_slots = ra.ring_slots ? std::min(std::max(ra.ring_slots, 2u), 16u) : 8u;
        while (!ra.ring_slots && plan.ring_slots > 2u && render_ring_bytes(ra, waves, plan.ring_slots) > budget) plan.ring_slots /= 2u;
    }
    plan.bytes = plan.ring_slots ? render_ring_bytes(ra, waves, plan.ring_slots) : 0;
    return plan;
}

// One frame's kernels on `stream` of the copy `d` (whose device is current): clears the counters and runs the render kernel
// into d_rgb; with sample chunks through d_chunks as `plan` says (slabs + the chunk-sum pass, or the waves' rings).  ev0 / ev1
// (nullable) are recorded around the render kernel.  Allocates nothing.
static int enqueue_render(const crt_scene *s, const DeviceScene &d, const crt_camera *cam, const crt_render_params *p, uint32_t chunks, float *d_rgb,
                          int *d_ids, unsigned long long *d_counters, float *d_chunks, const ChunkPlan &plan, hipEvent_t ev0, hipEvent_t ev1, hipStream_t stream,
                          const SplitStreams *split)
{
    const Tuning &t = s->tuning;
    RenderArgs ra = render_args(s, p, chunks);
    ra.chunk_mode = plan.mode;
    float *d_target = d_rgb;
    if (plan.mode == CM_RING) {
        ra.ring_lo = (uint32_t)(uintptr_t)d_chunks; ra.ring_hi = (uint32_t)((uint64_t)(uintptr_t)d_chunks >> 32);
        ra.ring_slots = plan.ring_slots;
        if (plan.fetch_units) ra.fetch_units = plan.fetch_units;
        if (plan.ring_batch) ra.ring_batch = plan.ring_batch;
    } else if (plan.mode == CM_SLAB) d_target = d_chunks; // raw chunk sums, [pixel][chunk][3]
    HIP_TRY(hipMemsetAsync(d_counters, 0, CRT_COUNTER_BYTES, stream));
    if (ev0) HIP_TRY(hipEventRecord(ev0, stream));
    HIP_TRY(launch_render(d.view, *reinterpret_cast<const CameraPOD *>(cam), ra, d_target, d_ids, d_counters, (p->flags & CRT_COUNT_WORK) != 0,
                          d.cus * (uint32_t)t.waves_per_cu, stream, t.waves_per_cu == 20 ? split : nullptr)); // (the split is 16 + 4 waves per CU)
    if (ev1) HIP_TRY(hipEventRecord(ev1, stream));
    if (plan.mode == CM_SLAB) HIP_TRY(launch_finalize_chunks(d_chunks, chunks, output_floats(p), p->spp, (p->flags & CRT_OUT_ACCUM) != 0, d_rgb, stream));
    return CRT_OK;
}

// The scratch set of `stream` on `d`, with counters, events and (sample chunks) slabs of the size this frame needs.
// counters, events and the fork / join events of a scratch set, once
static int init_scratch_set(DeviceScene &d, StreamScratch *x)
{
    if (x->d_counters) return CRT_OK;
    HIP_TRY(hipMalloc((void **)&x->d_counters, CRT_COUNTER_BYTES));
    HIP_TRY(hipEventCreate(&x->ev0));
    HIP_TRY(hipEventCreate(&x->ev1));
    HIP_TRY(hipEventCreate(&x->ev_done));
    HIP_TRY(make_split_streams(x->split, d.noalloc_split.helper)); // (the device's one helper stream, made at commit; own events)
    return CRT_OK;
}

static int acquire_scratch(const crt_scene *s, DeviceScene &d, hipStream_t stream, const crt_render_params *p, uint32_t chunks, StreamScratch **out)
{
    hipEvent_t handover = nullptr;
    StreamScratch *x = d.scratch_for(stream, &handover);
    if (handover) HIP_TRY(hipStreamWaitEvent(stream, handover, 0)); // the set's previous owner finishes first
    int rc = init_scratch_set(d, x);
    if (rc) return rc;
    x->plan = plan_chunks(s, d, p, chunks, (size_t)s->tuning.slab_budget_mb << 20, false);
    if (x->chunk_bytes < x->plan.bytes) {
        if (x->d_chunks) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(x->d_chunks); }
        x->d_chunks = nullptr; x->chunk_bytes = 0;
        HIP_TRY(hipMalloc((void **)&x->d_chunks, x->plan.bytes));
        x->chunk_bytes = x->plan.bytes;
    }
    *out = x;
    return CRT_OK;
}

// Synchronises `stream` and reads the counters of the frame last rendered with set `x`.
static int read_counters(StreamScratch *x, hipStream_t stream, unsigned long long c[32], float *kernel_ms)
{
    HIP_TRY(hipMemcpyAsync(c, x->d_counters, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (c[CRT_COUNTER_GUARD]) return fail(CRT_ERR_HIP, "crt_render: a render wave left through its trip-counter guard (scheduling bug): the frame is incomplete");
    HIP_TRY(hipEventElapsedTime(kernel_ms, x->ev0, x->ev1));
    return CRT_OK;
}

static void add_stats(crt_stats *stats, const unsigned long long c[32], float kernel_ms)
{
    stats->rays += c[0]; stats->nodes += c[1]; stats->tri_tests += c[2]; stats->sphere_tests += c[3]; stats->tie_refolds += c[4]; stats->shaded_hits += c[5];
    for (int k = 0; k < 10; k++) stats->sched[k] += c[8 + k];
    stats->kernel_ms = std::max(stats->kernel_ms, kernel_ms);
}

static void print_sections(const unsigned long long c[32])
{ // tuning aid: where the waves of the instrumented kernel spend their time
    static const char *const name[10] = { "votes", "hit+shade", "store+fetch", "philox", "scatter", "camera", "spheres+root box", "node steps", "leaf steps", "ring sums" };
    unsigned long long total = 0;
    for (int k = 0; k < 10; k++) total += c[18 + k];
    for (int k = 0; k < 10; k++) fprintf(stderr, "crt sections: %-17s %5.1f %%\n", name[k], total ? 100.0 * (double)c[18 + k] / (double)total : 0.0);
}

template <class T>
static int grow(T **buf, size_t *have, size_t need)
{
    if (*have >= need) return CRT_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *have = 0;
    HIP_TRY(hipMalloc((void **)buf, need * sizeof(T)));
    *have = need;
    return CRT_OK;
}

extern "C" int crt_render(const crt_scene *cs, const crt_camera *cam, const crt_render_params *p, float *out_rgb, int32_t *out_ids, crt_stats *stats)
{
    crt_scene *s = const_cast<crt_scene *>(cs);
    uint32_t chunks = 1;
    int rc = check_render_params(s, cam, p, out_rgb, "crt_render", &chunks);
    if (rc) return rc;
    DeviceScene &d = s->dev[0];
    HIP_TRY(hipSetDevice(d.device));
    hipStream_t stream = (hipStream_t)p->stream;
    const size_t npix = (size_t)p->width * p->height;
    const size_t rgb_floats = output_floats(p);
    StreamScratch *x = nullptr;
    std::unique_lock<std::mutex> lock(s->mutex); // held while the frame is enqueued, not while the host waits for it
    if ((rc = acquire_scratch(s, d, stream, p, chunks, &x))) return rc;

    const bool stated = (p->flags & CRT_OUT_DEVICE) != 0;
    const bool rgb_on_device = stated || is_device_pointer(out_rgb);
    const bool ids_on_device = out_ids ? (stated || is_device_pointer(out_ids)) : true;
    float *d_rgb = out_rgb;
    int *d_ids = out_ids;
    if (!rgb_on_device) { // staged through this stream's own buffer: host-pointer renders on different streams do not meet
        if (x->stage_rgb_floats < rgb_floats) HIP_TRY(hipStreamSynchronize(stream));
        if ((rc = grow(&x->d_stage_rgb, &x->stage_rgb_floats, rgb_floats))) return rc;
        d_rgb = x->d_stage_rgb;
        // a partial (tile_ranks > 1, row-major) render leaves other ranks' pixels untouched: start from the caller's content
        if (p->tile_ranks > 1 && !(p->flags & CRT_OUT_COMPACT)) HIP_TRY(hipMemcpyAsync(d_rgb, out_rgb, rgb_floats * sizeof(float), hipMemcpyHostToDevice, stream));
    }
    if (out_ids && !ids_on_device) {
        if (x->stage_ids < npix) HIP_TRY(hipStreamSynchronize(stream));
        if ((rc = grow(&x->d_stage_ids, &x->stage_ids, npix))) return rc;
        d_ids = x->d_stage_ids;
        if (p->tile_ranks > 1) HIP_TRY(hipMemcpyAsync(d_ids, out_ids, npix * sizeof(int), hipMemcpyHostToDevice, stream));
    }

    if ((rc = enqueue_render(s, d, cam, p, chunks, d_rgb, d_ids, x->d_counters, x->d_chunks, x->plan, x->ev0, x->ev1, stream, &x->split))) return rc;
    HIP_TRY(hipEventRecord(x->ev_done, stream));
    x->recorded = true;

    if (!rgb_on_device) HIP_TRY(hipMemcpyAsync(out_rgb, d_rgb, rgb_floats * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (out_ids && !ids_on_device) HIP_TRY(hipMemcpyAsync(out_ids, d_ids, npix * sizeof(int), hipMemcpyDeviceToHost, stream));
    lock.unlock();
    if (stats) {
        unsigned long long c[32];
        float ms = 0.0f;
        if ((rc = read_counters(x, stream, c, &ms))) return rc;
        memset(stats, 0, sizeof *stats);
        add_stats(stats, c, ms);
        if ((p->flags & CRT_COUNT_WORK) && s->tuning.print_sections) print_sections(c);
    } else if (!rgb_on_device || (out_ids && !ids_on_device)) {
        HIP_TRY(hipStreamSynchronize(stream)); // host buffers must be complete on return
    }
    return CRT_OK;
}

// crt_render for callers that cannot allocate (HIP graph capture, tight serving loops): the counters and the chunk slabs
// live in the device's copy of a static __device__ array (crt_kernels.hip), whose address was taken when the scene was
// committed to that device, as were the helper stream and the two events of a two-launch frame (launch_split: event record /
// stream wait only).  No hipMalloc / hipFree / stream or event creation / host synchronisation here.
extern "C" int crt_render_noalloc(const crt_scene *cs, const crt_camera *cam, const crt_render_params *p, float *out_rgb, int32_t *out_ids,
                                  uint64_t *out_rays)
{
    crt_scene *s = const_cast<crt_scene *>(cs);
    uint32_t chunks = 1;
    int rc = check_render_params(s, cam, p, out_rgb, "crt_render_noalloc", &chunks);
    if (rc) return rc;
    const DeviceScene &d = s->dev[0];
    unsigned char *scratch = (unsigned char *)noalloc_scratch_ptr(d.device);
    if (!scratch) return fail(CRT_ERR_HIP, "crt_render_noalloc: the static scratch of this device was not set up at commit");
    // sample chunks must fit the static scratch: as slabs, else as rings with fewer blocks in flight per wave if need be (at least two)
    const ChunkPlan plan = plan_chunks(s, d, p, chunks, (size_t)CRT_NOALLOC_SLAB_BYTES, true);
    if (plan.mode == CM_RING && plan.ring_slots == 0)
        return fail(CRT_ERR_LIMIT, "crt_render_noalloc: the sample-chunk ring (persistent waves x 2 blocks x units per block x 12 bytes; a block is at least one pixel's "
                                   "spp samples) exceeds the static scratch (CRT_NOALLOC_SLAB_BYTES = 32 MiB): use sample_chunks <= 1 (no scratch) or crt_render");
    HIP_TRY(hipSetDevice(d.device));
    hipStream_t stream = (hipStream_t)p->stream;
    unsigned long long *d_counters = (unsigned long long *)scratch;
    float *d_chunks = (float *)(scratch + CRT_NOALLOC_SLAB_OFFSET);
    if ((rc = enqueue_render(s, d, cam, p, chunks, out_rgb, out_ids, d_counters, d_chunks, plan, nullptr, nullptr, stream, &d.noalloc_split))) return rc; // (the second stream and its events were made at commit)
    if (out_rays) HIP_TRY(hipMemcpyAsync(out_rays, d_counters, sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
    return CRT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// multi-GPU: one process, n devices (crt_scene_commit_devices / crt_render_multi)
// ---------------------------------------------------------------------------------------------------------
namespace {
// RCCL is loaded on first use (dlopen): a single-GPU user of this library never maps it, and a process that already has
// an RCCL (torch's) keeps that one.  Only the entry points named here are used.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if (handle) return true;
        for (const char *name : { "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" }) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
        AllGather = (decltype(AllGather))dlsym(handle, "ncclAllGather");
        Reduce = (decltype(Reduce))dlsym(handle, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather && Reduce && GetErrorString;
    }
};
Rccl g_rccl;
} // namespace

#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(CRT_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

void RankWorkers::run(uint32_t r)
{
    uint64_t seen = 0;
    for (;;) {
        const std::function<int(uint32_t)> *j;
        {
            std::unique_lock<std::mutex> l(mu);
            wake.wait(l, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            j = job;
        }
        const int code = (*j)(r);
        std::lock_guard<std::mutex> l(mu);
        rc[r] = code;
        if (code) msg[r] = g_error; // (the error text is per thread: hand it to the caller's)
        if (--pending == 0) done.notify_one();
    }
}

int RankWorkers::run_all(const std::function<int(uint32_t)> &j)
{
    {
        std::lock_guard<std::mutex> l(mu);
        job = &j;
        pending = (uint32_t)threads.size();
        generation++;
    }
    wake.notify_all();
    const int rc0 = j(0);
    {
        std::unique_lock<std::mutex> l(mu);
        done.wait(l, [&] { return pending == 0; });
    }
    if (rc0) return rc0;
    for (size_t r = 1; r < rc.size(); r++)
        if (rc[r]) return fail(rc[r], msg[r]);
    return CRT_OK;
}

void crt_scene::release_multi()
{
    MultiState &m = multi;
    m.workers.reset(); // (no frame is being enqueued: the threads are idle)
    for (size_t r = 0; r < m.rank.size(); r++) {
        MultiRank &k = m.rank[r];
        if (r < dev.size() && dev[r].device >= 0) (void)hipSetDevice(dev[r].device);
        for (int i = 0; i < 2; i++) {
            if (k.stream[i]) { (void)hipStreamSynchronize(k.stream[i]); }
        }
        if (k.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)k.comm);
        for (int i = 0; i < 2; i++) {
            if (k.stream[i]) (void)hipStreamDestroy(k.stream[i]);
            if (k.d_local[i]) (void)hipFree(k.d_local[i]);
            if (k.d_all[i]) (void)hipFree(k.d_all[i]);
            if (k.sent[i]) (void)hipEventDestroy(k.sent[i]);
        }
    }
    if (!dev.empty() && dev[0].device >= 0) (void)hipSetDevice(dev[0].device);
    if (m.d_stage_rgb) (void)hipFree(m.d_stage_rgb);
    if (m.d_stage_ids) (void)hipFree(m.d_stage_ids);
    for (int i = 0; i < 2; i++) if (m.frame_done[i]) (void)hipEventDestroy(m.frame_done[i]);
    if (m.fork) (void)hipEventDestroy(m.fork);
    m = MultiState();
}

extern "C" int crt_scene_commit_devices(crt_scene *s, const int *devices, uint32_t n, uint32_t flags)
{
    if (!s || !devices) return fail(CRT_ERR_INVALID, "crt_scene_commit_devices: null argument");
    if (n == 0 || n > 64) return fail(CRT_ERR_INVALID, "crt_scene_commit_devices: 1 to 64 devices");
    int rc = prepare_host(s, devices[0], flags);
    if (rc) return rc;
    s->dev.resize(n);
    for (uint32_t k = 0; k < n; k++)
        if ((rc = upload_replica(s, s->dev[k], devices[k]))) { s->release_device(); return rc; }
    s->device = devices[0];
    s->committed = true;

    MultiState &m = s->multi;
    m.rank.resize(n);
    bool distinct = true;
    for (uint32_t a = 0; a < n; a++) for (uint32_t b = a + 1; b < n; b++) if (devices[a] == devices[b]) distinct = false;
    const char *tr = getenv("CRT_MULTI_TRANSPORT");
    m.use_rccl = distinct && n > 1;
    if (tr && !strcmp(tr, "copy")) m.use_rccl = false;
    if (tr && !strcmp(tr, "rccl")) {
        if (!distinct) { s->release_device(); return fail(CRT_ERR_INVALID, "crt_scene_commit_devices: CRT_MULTI_TRANSPORT=rccl needs distinct devices (a communicator refuses duplicates)"); }
        m.use_rccl = true;
    }
    auto bail = [&](int code) { s->release_device(); return code; };
    m.peer_ids = true;
    for (uint32_t r = 0; r < n; r++) {
        if (hipSetDevice(devices[r]) != hipSuccess) return bail(hip_fail(hipGetLastError(), "hipSetDevice"));
        for (int i = 0; i < 2; i++) {
            hipError_t e = hipStreamCreateWithFlags(&m.rank[r].stream[i], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&m.rank[r].sent[i], hipEventDisableTiming);
            if (e != hipSuccess) return bail(hip_fail(e, "crt_scene_commit_devices: stream / event creation"));
        }
        if (devices[r] != devices[0]) { // so that rank r can write its primary-hit ids straight into the root's buffer
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[r], devices[0]) != hipSuccess) { (void)hipGetLastError(); can = 0; }
            if (can) {
                const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
                (void)hipGetLastError();
            }
            if (!can) m.peer_ids = false;
        }
    }
    if (hipSetDevice(devices[0]) != hipSuccess) return bail(hip_fail(hipGetLastError(), "hipSetDevice"));
    for (int i = 0; i < 2; i++) {
        const hipError_t e = hipEventCreateWithFlags(&m.frame_done[i], hipEventDisableTiming);
        if (e != hipSuccess) return bail(hip_fail(e, "hipEventCreateWithFlags"));
    }
    {
        const hipError_t e = hipEventCreateWithFlags(&m.fork, hipEventDisableTiming);
        if (e != hipSuccess) return bail(hip_fail(e, "hipEventCreateWithFlags"));
    }
    if (m.use_rccl) {
        if (!g_rccl.load()) return bail(fail(CRT_ERR_HIP, "crt_scene_commit_devices: librccl.so.1 could not be loaded (set CRT_MULTI_TRANSPORT=copy for peer copies)"));
        std::vector<ncclComm_t> comms(n);
        const ncclResult_t r_ = g_rccl.CommInitAll(comms.data(), (int)n, devices);
        if (r_ != ncclSuccess) return bail(fail(CRT_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r_)));
        for (uint32_t r = 0; r < n; r++) m.rank[r].comm = comms[r];
    }
    // Rank workers when every rank has a device of its own (CRT_MULTI_THREADS=1 / 0 forces / forbids).  Logical ranks that share a GPU
    // -- the one-GPU rehearsal -- are issued by the caller's thread: eight threads feeding one device mostly wait for each other
    // inside the runtime (8 ranks on one MI355X: 0.46 -> 0.25 ms per call at best, median 1.15 -> 0.70), and the order in which the
    // ranks' launches reach that GPU then varies from frame to frame (9.9 -> 10.7 ms per frame).
    const int threads = get_env_int("CRT_MULTI_THREADS", -1);
    if (n > 1 && (threads > 0 || (threads < 0 && distinct))) m.workers.reset(new RankWorkers(n));
    m.ready = true;
    return CRT_OK;
}

extern "C" int crt_render_multi(const crt_scene *cs, const crt_camera *cam, const crt_render_params *p, int mode, float *out_rgb, int32_t *out_ids,
                                crt_stats *stats)
{
    crt_scene *s = const_cast<crt_scene *>(cs);
    uint32_t chunks = 1;
    int rc = check_render_params(s, cam, p, out_rgb, "crt_render_multi", &chunks);
    if (rc) return rc;
    MultiState &m = s->multi;
    if (!m.ready) return fail(CRT_ERR_INVALID, "crt_render_multi: commit the scene with crt_scene_commit_devices first");
    if (mode != CRT_MULTI_TILE && mode != CRT_MULTI_SAMPLE) return fail(CRT_ERR_INVALID, "crt_render_multi: unknown mode");
    if (p->tile_rank != 0 || p->tile_ranks != 1 || (p->flags & (CRT_OUT_ACCUM | CRT_OUT_COMPACT)))
        return fail(CRT_ERR_INVALID, "crt_render_multi: the library partitions the frame itself (tile_rank 0 of 1, no CRT_OUT_ACCUM / CRT_OUT_COMPACT)");
    const uint32_t n = (uint32_t)s->dev.size();
    if (mode == CRT_MULTI_SAMPLE && p->spp % n) return fail(CRT_ERR_INVALID, "crt_render_multi: spp must be a multiple of the number of ranks in sample mode");
    if (out_ids && !m.peer_ids) return fail(CRT_ERR_INVALID, "crt_render_multi: out_ids needs peer access from every rank's device to devices[0]");
    const int root = s->dev[0].device;
    const size_t npix = (size_t)p->width * p->height, frame_floats = npix * 3;

    // what every rank renders
    std::vector<crt_render_params> q(n, *p);
    for (uint32_t r = 0; r < n; r++) {
        if (mode == CRT_MULTI_TILE) { q[r].tile_rank = r; q[r].tile_ranks = n; q[r].flags |= CRT_OUT_COMPACT; }
        else {
            q[r].spp = p->spp / n; q[r].sample_offset = p->sample_offset + r * (p->spp / n); q[r].flags |= CRT_OUT_ACCUM;
            uint32_t c = chunks == p->spp ? q[r].spp : chunks; // one sample per unit stays one sample per unit
            while (c > 1 && q[r].spp % c) c--;                 // else the rank's share of the samples must still divide into the chunks
            q[r].sample_chunks = c;
        }
    }
    const uint32_t rank_chunks = q[0].sample_chunks > 1 ? q[0].sample_chunks : 1u;
    const size_t part = output_floats(&q[0]);                                  // floats of one rank's part (the same for every rank)
    const size_t all = (mode == CRT_MULTI_TILE || !m.use_rccl) ? part * n : part; // the root's assembly buffer (ncclReduce needs one part only)

    // buffers (grow only; a resize first waits for everything in flight)
    bool resize = false;
    for (uint32_t r = 0; r < n; r++) {
        if (m.rank[r].local_floats < part) resize = true;
        const bool needs_all = r == 0 || (m.use_rccl && mode == CRT_MULTI_TILE); // all-gather: every rank receives
        if (needs_all && m.rank[r].all_floats < all) resize = true;
    }
    if (resize) {
        for (uint32_t r = 0; r < n; r++) {
            HIP_TRY(hipSetDevice(s->dev[r].device));
            for (int i = 0; i < 2; i++) HIP_TRY(hipStreamSynchronize(m.rank[r].stream[i]));
        }
        for (uint32_t r = 0; r < n; r++) {
            HIP_TRY(hipSetDevice(s->dev[r].device));
            MultiRank &k = m.rank[r];
            if (k.local_floats < part) {
                for (int i = 0; i < 2; i++) { size_t have = k.local_floats; if ((rc = grow(&k.d_local[i], &have, part))) return rc; }
                k.local_floats = part;
            }
            const bool needs_all = r == 0 || (m.use_rccl && mode == CRT_MULTI_TILE);
            if (needs_all && k.all_floats < all) {
                for (int i = 0; i < 2; i++) { size_t have = k.all_floats; if ((rc = grow(&k.d_all[i], &have, all))) return rc; }
                k.all_floats = all;
            }
        }
    }

    HIP_TRY(hipSetDevice(root));
    hipStream_t user = (hipStream_t)p->stream;
    const bool stated = (p->flags & CRT_OUT_DEVICE) != 0;
    const bool rgb_on_device = stated || is_device_pointer(out_rgb);
    const bool ids_on_device = out_ids ? (stated || is_device_pointer(out_ids)) : true;
    float *d_rgb = out_rgb;
    int *d_ids = out_ids;
    if (!rgb_on_device) { if ((rc = grow(&m.d_stage_rgb, &m.stage_rgb_floats, frame_floats))) return rc; d_rgb = m.d_stage_rgb; }
    if (out_ids && !ids_on_device) { if ((rc = grow(&m.d_stage_ids, &m.stage_ids, npix))) return rc; d_ids = m.d_stage_ids; }

    const int set = (int)(m.frames & 1u);
    m.frames++;
    HIP_TRY(hipEventRecord(m.fork, user)); // what the caller has enqueued on its stream so far

    // ---- every rank renders its part on its own stream, enqueued by its own thread ----
    std::vector<StreamScratch *> x(n, nullptr);
    const std::function<int(uint32_t)> enqueue_rank = [&](uint32_t r) -> int {
        MultiRank &k = m.rank[r];
        HIP_TRY(hipSetDevice(s->dev[r].device));
        hipStream_t st = k.stream[set];
        // The renders read only the scene and write only the library's own buffers, so they do not wait for the caller's stream
        // (it carries the previous frame's completion: waiting for it would serialise the frames); the epilogue, which writes
        // out_rgb, does.  Ids go straight into the caller's buffer: then the renders wait as well.
        if (out_ids) HIP_TRY(hipStreamWaitEvent(st, m.fork, 0));
        // this set's previous frame (two frames ago): its parts on the root have been consumed once that frame was complete
        if (m.done_recorded[set]) HIP_TRY(hipStreamWaitEvent(st, m.frame_done[set], 0));
        int rc_ = acquire_scratch(s, s->dev[r], st, &q[r], rank_chunks, &x[r]);
        if (rc_) return rc_;
        q[r].stream = st;
        if ((rc_ = enqueue_render(s, s->dev[r], cam, &q[r], rank_chunks, k.d_local[set], d_ids, x[r]->d_counters, x[r]->d_chunks, x[r]->plan, x[r]->ev0, x[r]->ev1, st, &x[r]->split)))
            return rc_;
        HIP_TRY(hipEventRecord(x[r]->ev_done, st));
        x[r]->recorded = true;
        return CRT_OK;
    };
    if (m.workers) rc = m.workers->run_all(enqueue_rank);
    else for (uint32_t r = 0; r < n && !rc; r++) rc = enqueue_rank(r);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(root));

    // ---- ONE exchange: the parts travel to the root ----
    float *d_all = m.rank[0].d_all[set];
    if (m.use_rccl) {
        RCCL_TRY(g_rccl.GroupStart());
        for (uint32_t r = 0; r < n; r++) {
            MultiRank &k = m.rank[r];
            ncclResult_t e;
            if (mode == CRT_MULTI_TILE) e = g_rccl.AllGather(k.d_local[set], k.d_all[set], part, ncclFloat, (ncclComm_t)k.comm, k.stream[set]);
            else e = g_rccl.Reduce(k.d_local[set], r == 0 ? (void *)d_all : (void *)k.d_local[set], part, ncclFloat, ncclSum, 0, (ncclComm_t)k.comm, k.stream[set]);
            if (e != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(CRT_ERR_HIP, std::string("RCCL collective: ") + g_rccl.GetErrorString(e)); }
        }
        RCCL_TRY(g_rccl.GroupEnd());
        HIP_TRY(hipSetDevice(root));
    } else {
        for (uint32_t r = 0; r < n; r++) {
            MultiRank &k = m.rank[r];
            HIP_TRY(hipSetDevice(s->dev[r].device));
            HIP_TRY(hipMemcpyPeerAsync(d_all + (size_t)r * part, root, k.d_local[set], s->dev[r].device, part * sizeof(float), k.stream[set]));
            if (r > 0) { HIP_TRY(hipEventRecord(k.sent[set], k.stream[set])); k.sent_recorded[set] = true; }
        }
        HIP_TRY(hipSetDevice(root));
        for (uint32_t r = 1; r < n; r++) HIP_TRY(hipStreamWaitEvent(m.rank[0].stream[set], m.rank[r].sent[set], 0));
    }

    // ---- epilogue on the root, behind the previous frame's (frames complete in submission order) ----
    hipStream_t rs = m.rank[0].stream[set];
    HIP_TRY(hipStreamWaitEvent(rs, m.fork, 0)); // what the caller enqueued before this call (the last reader of out_rgb) comes first
    if (m.done_recorded[set ^ 1]) HIP_TRY(hipStreamWaitEvent(rs, m.frame_done[set ^ 1], 0));
    if (mode == CRT_MULTI_TILE) HIP_TRY(launch_untile(d_all, p->width, p->height, n, d_rgb, rs));
    else if (m.use_rccl) HIP_TRY(launch_finalize(d_all, frame_floats, p->spp, d_rgb, rs));
    else HIP_TRY(launch_sum_frames(d_all, n, frame_floats, p->spp, false, d_rgb, rs)); // the n sums added in rank order, /spp, gamma
    if (out_ids && n > 1) { // other ranks wrote ids into the root's buffer from their own streams
        for (uint32_t r = 1; r < n; r++) {
            if (m.use_rccl) { // (copy transport: sent[r] already covers rank r's render)
                HIP_TRY(hipSetDevice(s->dev[r].device));
                HIP_TRY(hipEventRecord(m.rank[r].sent[set], m.rank[r].stream[set]));
                HIP_TRY(hipSetDevice(root));
                HIP_TRY(hipStreamWaitEvent(rs, m.rank[r].sent[set], 0));
            }
        }
    }
    HIP_TRY(hipEventRecord(m.frame_done[set], rs));
    m.done_recorded[set] = true;
    HIP_TRY(hipStreamWaitEvent(user, m.frame_done[set], 0));

    if (!rgb_on_device) HIP_TRY(hipMemcpyAsync(out_rgb, d_rgb, frame_floats * sizeof(float), hipMemcpyDeviceToHost, user));
    if (out_ids && !ids_on_device) HIP_TRY(hipMemcpyAsync(out_ids, d_ids, npix * sizeof(int), hipMemcpyDeviceToHost, user));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (uint32_t r = 0; r < n; r++) {
            HIP_TRY(hipSetDevice(s->dev[r].device));
            unsigned long long c[32];
            float ms = 0.0f;
            if ((rc = read_counters(x[r], m.rank[r].stream[set], c, &ms))) return rc;
            add_stats(stats, c, ms);
        }
        HIP_TRY(hipSetDevice(root));
        HIP_TRY(hipStreamSynchronize(user));
    } else if (!rgb_on_device || (out_ids && !ids_on_device)) {
        HIP_TRY(hipStreamSynchronize(user));
    }
    return CRT_OK;
}

extern "C" int crt_untile(const float *compact_all, uint32_t width, uint32_t height, uint32_t tile_ranks, float *out_rgb, int device, void *stream)
{
    if (!compact_all || !out_rgb || tile_ranks == 0 || width == 0 || height == 0) return fail(CRT_ERR_INVALID, "crt_untile: bad argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(launch_untile(compact_all, width, height, tile_ranks, out_rgb, (hipStream_t)stream));
    return CRT_OK;
}

extern "C" int crt_finalize(const float *accum, uint32_t width, uint32_t height, uint32_t total_spp, float *out_rgb, int device, void *stream)
{
    if (!accum || !out_rgb || total_spp == 0 || width == 0 || height == 0) return fail(CRT_ERR_INVALID, "crt_finalize: bad argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(launch_finalize(accum, (size_t)width * height * 3, total_spp, out_rgb, (hipStream_t)stream));
    return CRT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// output path (row N2): 8-bit quantisation on the device, PNG container on the host
// ---------------------------------------------------------------------------------------------------------
extern "C" int crt_quantize_rgb8(const float *rgb, uint32_t width, uint32_t height, uint8_t *out_rgb8, int device, void *stream)
{
    if (!rgb || !out_rgb8 || width == 0 || height == 0) return fail(CRT_ERR_INVALID, "crt_quantize_rgb8: bad argument");
    if (device < 0) return fail(CRT_ERR_NO_DEVICE, "crt_quantize_rgb8: needs a HIP device (no host path)");
    HIP_TRY(hipSetDevice(device));
    const size_t n = (size_t)width * height * 3;
    const bool in_dev = is_device_pointer(rgb), out_dev = is_device_pointer(out_rgb8);
    float *d_in = const_cast<float *>(rgb);
    uint8_t *d_out = out_rgb8;
    int rc = CRT_OK;
    // host buffers are staged through the device (as crt_render does): the conversion itself always runs on the GPU
    if (!in_dev) {
        HIP_TRY(hipMalloc((void **)&d_in, n * sizeof(float)));
        if (hipMemcpyAsync(d_in, rgb, n * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) rc = CRT_ERR_HIP;
    }
    if (rc == CRT_OK && !out_dev && hipMalloc((void **)&d_out, n) != hipSuccess) { d_out = nullptr; rc = CRT_ERR_HIP; }
    if (rc == CRT_OK && launch_quantize(d_in, n, d_out, (hipStream_t)stream) != hipSuccess) rc = CRT_ERR_HIP;
    if (rc == CRT_OK && !out_dev && hipMemcpyAsync(out_rgb8, d_out, n, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) rc = CRT_ERR_HIP;
    if (!in_dev || !out_dev) {
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = CRT_ERR_HIP;
        if (!in_dev && d_in) (void)hipFree(d_in);
        if (!out_dev && d_out) (void)hipFree(d_out);
    }
    if (rc != CRT_OK) return fail(rc, "crt_quantize_rgb8: HIP call failed");
    return CRT_OK;
}

namespace {
struct PngOut {
    FILE *f;
    uint32_t crc;
    static const uint32_t *table()
    {
        static uint32_t t[256];
        static bool ready = false;
        if (!ready) {
            for (uint32_t n = 0; n < 256; n++) {
                uint32_t c = n;
                for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
                t[n] = c;
            }
            ready = true;
        }
        return t;
    }
    bool put(const void *p, size_t n)
    {
        const unsigned char *b = (const unsigned char *)p;
        const uint32_t *t = table();
        for (size_t k = 0; k < n; k++) crc = t[(crc ^ b[k]) & 0xFFu] ^ (crc >> 8);
        return fwrite(p, 1, n, f) == n;
    }
    bool be32(uint32_t v)
    {
        const unsigned char b[4] = { (unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v };
        return put(b, 4);
    }
    bool begin(const char type[4], uint32_t length)
    {
        const unsigned char b[4] = { (unsigned char)(length >> 24), (unsigned char)(length >> 16), (unsigned char)(length >> 8), (unsigned char)length };
        if (fwrite(b, 1, 4, f) != 4) return false;
        crc = 0xFFFFFFFFu;
        return put(type, 4);
    }
    bool end()
    {
        const uint32_t v = crc ^ 0xFFFFFFFFu;
        const unsigned char b[4] = { (unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v };
        return fwrite(b, 1, 4, f) == 4;
    }
};
} // namespace

extern "C" int crt_write_png(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height)
{
    if (!path || !rgb8 || width == 0 || height == 0) return fail(CRT_ERR_INVALID, "crt_write_png: bad argument");
    const uint64_t row = 1ull + 3ull * width, raw = row * height; // filter byte + pixels
    const uint64_t nblocks = (raw + 65534) / 65535;
    const uint64_t idat = 2 + raw + 5 * nblocks + 4;
    if (idat > 0x7FFFFFFFull) return fail(CRT_ERR_LIMIT, "crt_write_png: image too large for one IDAT chunk");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(CRT_ERR_IO, "crt_write_png: cannot open output file");
    PngOut o{ f, 0 };
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    bool ok = fwrite(sig, 1, 8, f) == 8;
    const unsigned char ihdr_tail[5] = { 8, 2, 0, 0, 0 }; // bit depth 8, colour type 2 (RGB), deflate, filter method 0, no interlace
    ok = ok && o.begin("IHDR", 13) && o.be32(width) && o.be32(height) && o.put(ihdr_tail, 5) && o.end();
    ok = ok && o.begin("IDAT", (uint32_t)idat);
    const unsigned char zhdr[2] = { 0x78, 0x01 };
    ok = ok && o.put(zhdr, 2);
    // zlib stream of stored blocks over the filtered scanlines; Adler-32 of the same bytes
    uint32_t a = 1, b = 0;
    std::vector<unsigned char> buf;
    buf.reserve(65535);
    uint64_t produced = 0;
    auto flush = [&](bool last) {
        const uint16_t len = (uint16_t)buf.size(), nlen = (uint16_t)~len;
        const unsigned char h[5] = { (unsigned char)(last ? 1 : 0), (unsigned char)(len & 0xFF), (unsigned char)(len >> 8), (unsigned char)(nlen & 0xFF),
                                     (unsigned char)(nlen >> 8) };
        ok = ok && o.put(h, 5) && (buf.empty() || o.put(buf.data(), buf.size()));
        buf.clear();
    };
    auto emit = [&](const unsigned char *p, size_t n) {
        for (size_t k = 0; k < n; k++) { a = (a + p[k]) % 65521u; b = (b + a) % 65521u; }
        while (n) {
            const size_t room = 65535 - buf.size(), take = n < room ? n : room;
            buf.insert(buf.end(), p, p + take);
            p += take; n -= take; produced += take;
            if (buf.size() == 65535) flush(produced == raw);
        }
    };
    for (uint32_t y = 0; y < height && ok; y++) {
        const unsigned char filter0 = 0;
        emit(&filter0, 1);
        emit(rgb8 + (size_t)y * 3 * width, (size_t)3 * width);
    }
    if (!buf.empty()) flush(true);
    ok = ok && o.be32((b << 16) | a) && o.end();
    ok = ok && o.begin("IEND", 0) && o.end();
    if (fclose(f) != 0) ok = false;
    if (!ok) return fail(CRT_ERR_IO, "crt_write_png: write failed");
    return CRT_OK;
}
