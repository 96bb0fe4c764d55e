// tests/hostsim/hostsim.cpp -- TEST-ONLY host build of the product's per-lane code (crt_core.h / crt_traverse.h) and
// host scene preparation (crt_api.cpp / crt_bvh.cpp).  It lets the CPU test suite single-step the exact traversal /
// tie-fold / shading logic the gfx950 kernel runs, one lane at a time, and compare it with the oracle before any GPU
// is involved.  It is never loaded by the cudaraytracer_amd package and is not a rendering fallback: the kernel
// launchers are stubbed to fail.
#include "../../cudaraytracer_amd/csrc/crt_api.cpp"

namespace crt {
hipError_t launch_render(const SceneView &, const CameraPOD &, const RenderArgs &, float *, int *, unsigned long long *, bool, uint32_t, hipStream_t, const SplitStreams *)
{
    return hipErrorNoDevice;
}
hipError_t launch_untile(const float *, uint32_t, uint32_t, uint32_t, float *, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_finalize(const float *, size_t, uint32_t, float *, hipStream_t) { return hipErrorNoDevice; }
hipError_t lbvh_build(const float *, const int *, uint32_t, const Box &, bool, const LbvhSphereInput *, uint32_t, uint32_t, uint32_t, LbvhOutput &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_quantize(const float *, size_t, uint8_t *, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_finalize_chunks(const float *, uint32_t, size_t, uint32_t, bool, float *, hipStream_t) { return hipErrorNoDevice; }
size_t render_ring_bytes(const RenderArgs &, uint32_t, uint32_t) { return 0; }
uint32_t render_ring_slots_that_fit(const RenderArgs &, uint32_t, size_t) { return 0; }
bool pool_kernel_built() { return false; }
hipError_t launch_sum_frames(const float *, uint32_t, size_t, uint32_t, bool, float *, hipStream_t) { return hipErrorNoDevice; }
hipError_t prepare_split_kernels() { return hipErrorNoDevice; }
hipError_t noalloc_scratch_init(int) { return hipErrorNoDevice; }
void *noalloc_scratch_ptr(int) { return nullptr; }
} // namespace crt

namespace {
struct ArrayStack {
    int v[CRT_STACK_DEPTH];
    int sp = 0;
    int max_sp = 0;
    void reset() { sp = 0; }
    void push(int x) { if (sp < CRT_STACK_DEPTH) v[sp] = x; sp++; if (sp > max_sp) max_sp = sp; }
    int pop() { if (sp == 0) return CRT_BVH_NONE; sp--; return v[sp]; }
    // the kernel's branch-free forms
    int peek() const { return sp == 0 ? CRT_BVH_NONE : v[sp - 1]; }
    void push_if(int x, bool c) { if (c) push(x); }
    void pop_if(bool c) { if (c && sp > 0) sp--; }
};
struct HostCounters {
    static constexpr bool sphere_slots = true;
    uint64_t nodes = 0, tris = 0, spheres = 0, rays = 0, refolds = 0, shades = 0;
    void node() { nodes++; }
    void tri() { tris++; }
    void sphere() { spheres++; }
    void ray() { rays++; }
    void refold() { refolds++; }
    void shade() { shades++; }
};
SceneView host_view(const crt_scene *s, std::vector<Sphere> &spheres)
{
    spheres = s->flat_spheres;
    SceneView v{};
    v.nodes = s->bvh.nodes.data(); v.tris = s->bvh.tris.data(); v.spheres = spheres.data(); v.shades = s->shades.data(); v.normals = s->normals.data();
    v.root_ref = s->bvh.root_ref; v.wide = s->bvh.wide ? 1u : 0u; v.nnodes = (uint32_t)(s->bvh.nodes.size() / (s->bvh.wide ? 8 : 4)); v.ntris = (uint32_t)(s->bvh.tris.size() / 3); v.nspheres = (uint32_t)spheres.size(); v.sphere_slots = 0;
    if (s->sphere_slots) std::memcpy(&v.sphere_slots, &s->sphere_slot_rmax, 4); // (the host counters compile the sphere-slot code in; this switches its pruning slack as on the device)
    v.root_min = s->bvh.root_box.mn; v.root_max = s->bvh.root_box.mx;
    return v;
}
} // namespace

// Same contract as crt_render (row-major output, host pointers), executed lane by lane on the CPU.
// stats_out[0..6] = rays, nodes, tri tests, sphere tests, tie refolds, deepest stack use, shaded hits.
extern "C" int hostsim_render(const crt_scene *s, const crt_camera *cam, const crt_render_params *p, float *out_rgb, int32_t *out_ids,
                              uint64_t *stats_out)
{
    if (!s || !s->prepared || !cam || !p || !out_rgb) return CRT_ERR_INVALID;
    std::vector<Sphere> spheres;
    const SceneView sc = host_view(s, spheres);
    const CameraPOD &c = *reinterpret_cast<const CameraPOD *>(cam);
    HostCounters cnt;
    ArrayStack stack;
    const uint32_t tiles_x = (p->width + 7u) / 8u, tiles_y = (p->height + 7u) / 8u;
    for (uint32_t tile = p->tile_rank; tile < tiles_x * tiles_y; tile += p->tile_ranks) {
        const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
        for (uint32_t lane = 0; lane < 64; lane++) {
            const uint32_t i = tx * 8u + (lane & 7u), j = ty * 8u + (lane >> 3);
            if (i >= p->width || j >= p->height) continue;
            const uint32_t pixel = j * p->width + i;
            f3 col = mk3(0.0f, 0.0f, 0.0f);
            for (uint32_t k = 0; k < p->spp; k++) {
                const uint32_t sample = p->sample_offset + k;
                int prim;
                const f3 r = sample_radiance(sc, c, i, j, pixel, sample, p->max_depth, (uint32_t)p->seed, (uint32_t)(p->seed >> 32), stack, cnt, prim);
                col = col + r;
                if (out_ids && sample == p->ids_sample) out_ids[pixel] = prim;
            }
            f3 o = col;
            if (!(p->flags & CRT_OUT_ACCUM)) {
                o = div3(col, (float)p->spp);
                o.x = o.x > 0 ? sqrtf(o.x) : 0.0f; o.y = o.y > 0 ? sqrtf(o.y) : 0.0f; o.z = o.z > 0 ? sqrtf(o.z) : 0.0f;
            }
            out_rgb[3 * (size_t)pixel] = o.x; out_rgb[3 * (size_t)pixel + 1] = o.y; out_rgb[3 * (size_t)pixel + 2] = o.z;
        }
    }
    if (stats_out) {
        stats_out[0] = cnt.rays; stats_out[1] = cnt.nodes; stats_out[2] = cnt.tris; stats_out[3] = cnt.spheres; stats_out[4] = cnt.refolds;
        stats_out[5] = (uint64_t)stack.max_sp;
        stats_out[6] = cnt.shades;
    }
    return CRT_OK;
}

// one ray through the product traversal: returns primitive id (-1 miss), writes t; force_tie_path != 0 runs the
// reference-order fold even when the fast path saw no near tie (the two must agree).
extern "C" int hostsim_trace(const crt_scene *s, const float o_[3], const float d_[3], float tmin, float *t_out, int force_tie_path, int *took_tie_path)
{
    if (!s || !s->prepared) return -3;
    std::vector<Sphere> spheres;
    const SceneView sc = host_view(s, spheres);
    HostCounters cnt;
    ArrayStack stack;
    const f3 o = mk3(o_[0], o_[1], o_[2]), d = mk3(d_[0], d_[1], d_[2]);
    float t = 0.0f;
    int code, mat = 0;
    if (!force_tie_path) {
        code = closest_hit(sc, o, d, tmin, FLT_MAX, t, mat, stack, cnt);
    } else {
        Search q; q.t = FLT_MAX; q.code = -1; q.mat = 0; q.rank = 0; q.near_tie = false;
        walk<MODE_CLOSEST>(sc, o, d, tmin, q, 0, 0.0f, stack, cnt);
        code = -1;
        if (q.code >= 0) {
            cnt.refold();
            code = refold(sc, o, d, tmin, FLT_MAX, q.t, t, mat, stack, cnt);
        }
    }
    if (took_tie_path) *took_tie_path = cnt.refolds > 0;
    if (code < 0) return -1;
    if (t_out) *t_out = t;
    return surface_at(sc, code, o, d, t).prim;
}

// ---------------------------------------------------------------------------------------------------------------------
// Walk traces for the schedule simulator (tools/sched_sim.cpp): the per-lane event sequence of every path segment of a
// subsample of the frame's 8x8 tiles, written as a byte stream:
//   0xFD pixel start, 0xFE sample (path) start, 0xFF path segment start, 0x01 one inner-node step, 0x10 + k one leaf step with
//   k primitive tests.  (Flat-sphere tests and shading are the same for every segment and are not recorded.)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
std::vector<unsigned char> *g_trace = nullptr;
int g_leaf_tests = 0;
struct TraceCounters {
    static constexpr bool sphere_slots = true;
    void node() { g_trace->push_back(0x01); }
    void tri() { g_leaf_tests++; }
    void sphere() {}
    void ray() { g_trace->push_back(0xFF); }
    void refold() {}
    void shade() {}
};
struct TraceStack : ArrayStack {
    int pop() { g_trace->push_back((unsigned char)(0x10 + (g_leaf_tests > 15 ? 15 : g_leaf_tests))); g_leaf_tests = 0; return ArrayStack::pop(); } // the end of every leaf step
};
} // namespace

extern "C" int hostsim_trace_file(const crt_scene *s, const crt_camera *cam, const crt_render_params *p, uint32_t tile_step, const char *path)
{
    if (!s || !s->prepared || !cam || !p || !path || tile_step == 0) return CRT_ERR_INVALID;
    std::vector<Sphere> spheres;
    const SceneView sc = host_view(s, spheres);
    const CameraPOD &c = *reinterpret_cast<const CameraPOD *>(cam);
    std::vector<unsigned char> trace;
    g_trace = &trace;
    TraceCounters cnt;
    TraceStack stack;
    const uint32_t tiles_x = (p->width + 7u) / 8u, tiles_y = (p->height + 7u) / 8u;
    for (uint32_t tile = 0; tile < tiles_x * tiles_y; tile += tile_step) {
        const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
        for (uint32_t lane = 0; lane < 64; lane++) {
            const uint32_t i = tx * 8u + (lane & 7u), j = ty * 8u + (lane >> 3);
            if (i >= p->width || j >= p->height) continue;
            trace.push_back(0xFD);
            for (uint32_t k = 0; k < p->spp; k++) {
                trace.push_back(0xFE);
                int prim;
                g_leaf_tests = 0;
                (void)sample_radiance(sc, c, i, j, j * p->width + i, p->sample_offset + k, p->max_depth, (uint32_t)p->seed, (uint32_t)(p->seed >> 32), stack, cnt, prim);
            }
        }
    }
    g_trace = nullptr;
    FILE *f = fopen(path, "wb");
    if (!f) return CRT_ERR_IO;
    const bool ok = fwrite(trace.data(), 1, trace.size(), f) == trace.size();
    fclose(f);
    return ok ? CRT_OK : CRT_ERR_IO;
}

// ---------------------------------------------------------------------------------------------------------------------
// Study: how much of the primary rays' traversal a per-pixel ENTRY POINT would save.  For every
// sampled pixel: the inner nodes its spp primary rays visit; the entry point is the deepest node such that every visited
// node lies on the path root -> entry or below the entry (no ray of the pixel enters a sibling on the way down).  Starting
// the pixel's primary walks there skips depth(entry) node steps per walking ray; rays that miss the root box pay one step.
// out[0] primary rays, [1] of them walking, [2] their node steps, [3] node steps saved, [4] steps added (non-walking rays of
// pixels with an entry point), [5] pixels, [6] pixels without any walking ray, [7] sum of entry depths over pixels with one.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int hostsim_entry_point_study(const crt_scene *s, const crt_camera *cam, const crt_render_params *p, uint32_t tile_step, uint64_t *out)
{
    if (!s || !s->prepared || !cam || !p || !out || tile_step == 0) return CRT_ERR_INVALID;
    std::vector<Sphere> spheres;
    const SceneView sc = host_view(s, spheres);
    const CameraPOD &c = *reinterpret_cast<const CameraPOD *>(cam);
    if (sc.wide) return CRT_ERR_INVALID; // (a study of the two-child format)
    const uint32_t nn = sc.nnodes;
    std::vector<int> parent(nn, -1), depth(nn, 0);
    for (uint32_t n = 0; n < nn; n++) // breadth-first numbering: parents come before children
        for (int k = 0; k < 2; k++) {
            const int ref = f2i(sc.nodes[4 * n + k].w);
            if (is_inner(ref)) { parent[ref] = (int)n; depth[ref] = depth[n] + 1; }
        }
    for (int k = 0; k < 8; k++) out[k] = 0;
    HostCounters cnt;
    ArrayStack stack;
    std::vector<uint32_t> mark(nn, 0);
    uint32_t stamp = 0;
    const uint32_t tiles_x = (p->width + 7u) / 8u, tiles_y = (p->height + 7u) / 8u;
    for (uint32_t tile = 0; tile < tiles_x * tiles_y; tile += tile_step) {
        const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
        for (uint32_t lane = 0; lane < 64; lane++) {
            const uint32_t i = tx * 8u + (lane & 7u), j = ty * 8u + (lane >> 3);
            if (i >= p->width || j >= p->height) continue;
            stamp++;
            std::vector<int> visited;
            uint64_t walking = 0, steps = 0;
            for (uint32_t k = 0; k < p->spp; k++) {
                f3 o, d;
                camera_ray(c, i, j, j * p->width + i, p->sample_offset + k, (uint32_t)p->seed, (uint32_t)(p->seed >> 32), o, d);
                Search q; q.t = FLT_MAX; q.code = -1; q.mat = 0; q.rank = 0; q.near_tie = false;
                Trav tv;
                trav_begin<MODE_CLOSEST>(sc, o, d, 0.001f, q, 0, 0.0f, tv, stack, cnt);
                if (tv.cur != CRT_BVH_NONE) walking++;
                while (tv.cur != CRT_BVH_NONE) {
                    while (is_inner(tv.cur)) {
                        if (mark[tv.cur] != stamp) { mark[tv.cur] = stamp; visited.push_back(tv.cur); }
                        steps++;
                        trav_inner(sc, tv, stack, cnt);
                    }
                    if (tv.cur < 0) trav_leaf<MODE_CLOSEST>(sc, o, d, 0.001f, q, 0, 0.0f, tv, stack, cnt);
                }
            }
            out[0] += p->spp; out[1] += walking; out[2] += steps; out[5]++;
            if (!walking) { out[6]++; continue; }
            // entry point: go down from the root while exactly one inner child was visited and ... (a leaf child may have been
            // entered too: then the walk must start here)
            int ep = sc.root_ref;
            for (;;) {
                int visited_children = 0, next = -1;
                bool leaf_child = false;
                for (int k = 0; k < 2; k++) {
                    const int ref = f2i(sc.nodes[4 * (size_t)ep + k].w);
                    if (is_inner(ref)) { if (mark[ref] == stamp) { visited_children++; next = ref; } }
                    else leaf_child = true; // cannot tell from the marks whether the leaf was entered: stay conservative
                }
                if (visited_children != 1 || leaf_child) break;
                ep = next;
            }
            out[3] += walking * (uint64_t)depth[ep];
            out[4] += depth[ep] ? (p->spp - walking) : 0;
            out[7] += (uint64_t)depth[ep];
        }
    }
    return CRT_OK;
}

// crt_core.h: div3_exact with the reciprocal commit would store for divisor `s` (crt_bvh.cpp: reciprocal_if_exact), n
// numerator triples; *y_out = that reciprocal (0: the divisor gets none and div3_exact divides the ordinary way)
extern "C" void hostsim_div3_exact(float s, const float *a, uint32_t n, float *q, float *y_out)
{
    const float y = crt::reciprocal_if_exact(s);
    *y_out = y;
    for (uint32_t k = 0; k < n; k++) {
        const crt::f3 r = crt::div3_exact(crt::mk3(a[3 * k], a[3 * k + 1], a[3 * k + 2]), s, y);
        q[3 * k] = r.x; q[3 * k + 1] = r.y; q[3 * k + 2] = r.z;
    }
}

// crt_core.h: sphere_t decides most sphere hits' box test from the sphere arithmetic (sphere_hit_implies_box) and evaluates the
// reference's literal slab test (reference_box_test) for the rest.  Audit of that shortcut over n rays against one sphere:
// out[0] = hits (root >= tmin), [1] = of them decided by the shortcut, [2] = of them passing the literal test, [3] = VIOLATIONS:
// the shortcut said "implied" and the literal test fails (must be 0).
extern "C" void hostsim_sphere_box_audit(const float center_[3], float radius_arg, const float *o_, const float *d_, uint64_t n, float tmin, uint64_t *out)
{
    using namespace crt;
    const f3 center = mk3(center_[0], center_[1], center_[2]);
    const float radius = sphere_radius(radius_arg), h = __builtin_fabsf(radius_arg);
    const f3 lo = mk3(center.x - h, center.y - h, center.z - h), hi = mk3(center.x + h, center.y + h, center.z + h);
    out[0] = out[1] = out[2] = out[3] = 0;
    for (uint64_t k = 0; k < n; k++) {
        const f3 o = mk3(o_[3 * k], o_[3 * k + 1], o_[3 * k + 2]), d = mk3(d_[3 * k], d_[3 * k + 1], d_[3 * k + 2]);
        const f3 oc = center - o;
        const float a = dot(d, d), b = dot(d, oc), q = dot(oc, oc), c = fma1(-radius, radius, q), disc = fma1(b, b, -(a * c));
        if (disc < 0) continue;
        const float sd = __builtin_sqrtf(disc);
        float root = (b - sd) / a;
        if (!(root >= tmin)) { root = (b + sd) / a; if (!(root >= tmin)) continue; }
        const bool implied = sphere_hit_implies_box(sphere_sure_term(center, radius), a, b, q, c, disc) || strictly_inside_box(center, h, o), literal = reference_box_test(lo, hi, o, d);
        out[0]++; out[1] += implied; out[2] += literal; out[3] += implied && !literal;
    }
}
