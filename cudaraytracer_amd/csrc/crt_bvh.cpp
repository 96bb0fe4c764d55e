// crt_bvh.cpp -- see crt_bvh.h.  Host code; float arithmetic that feeds ranks is written exactly as the
// reference evaluates it (no contraction: this file is built with -ffp-contract=off).
#include "crt_bvh.h"
#include "crt_lbvh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <thread>

namespace crt {

// ---------------------------------------------------------------------------------------------------------
// Reference boxes and Morton ranks
// ---------------------------------------------------------------------------------------------------------
static inline float min3f(float a, float b, float c) { return fminf(fminf(fminf(a, a), b), c); } // utility.h:45-51
static inline float max3f(float a, float b, float c) { return fmaxf(fmaxf(fmaxf(a, a), b), c); } // utility.h:53-59

// aabb.h:12-16: per axis the smaller coordinate becomes the interval minimum
static Box box_from_points(f3 p, f3 q)
{
    Box r;
    if (p.x < q.x) { r.mn.x = p.x; r.mx.x = q.x; } else { r.mn.x = q.x; r.mx.x = p.x; }
    if (p.y < q.y) { r.mn.y = p.y; r.mx.y = q.y; } else { r.mn.y = q.y; r.mx.y = p.y; }
    if (p.z < q.z) { r.mn.z = p.z; r.mx.z = q.z; } else { r.mn.z = q.z; r.mx.z = p.z; }
    return r;
}

Box reference_triangle_box(const HostTriangle &t)
{
    const float EPSILON = 1e-4f;
    return box_from_points(mk3(min3f(t.a.x, t.b.x, t.c.x) - EPSILON, min3f(t.a.y, t.b.y, t.c.y) - EPSILON, min3f(t.a.z, t.b.z, t.c.z) - EPSILON),
                           mk3(max3f(t.a.x, t.b.x, t.c.x) + EPSILON, max3f(t.a.y, t.b.y, t.c.y) + EPSILON, max3f(t.a.z, t.b.z, t.c.z) + EPSILON));
}

Box reference_sphere_box(const HostSphere &s)
{
    const f3 rvec = mk3(s.radius_arg, s.radius_arg, s.radius_arg);
    return box_from_points(s.center - rvec, s.center + rvec);
}

// interval.h:11-14 applied per axis (aabb.h:17-21)
static Box box_union(const Box &a, const Box &b)
{
    Box r;
    r.mn.x = a.mn.x <= b.mn.x ? a.mn.x : b.mn.x; r.mx.x = a.mx.x >= b.mx.x ? a.mx.x : b.mx.x;
    r.mn.y = a.mn.y <= b.mn.y ? a.mn.y : b.mn.y; r.mx.y = a.mx.y >= b.mx.y ? a.mx.y : b.mx.y;
    r.mn.z = a.mn.z <= b.mn.z ? a.mn.z : b.mn.z; r.mx.z = a.mx.z >= b.mx.z ? a.mx.z : b.mx.z;
    return r;
}

// bvh.h:13-20
static inline uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
// bvh.h:24-33
static inline uint32_t morton_3d(f3 p)
{
    const float x = fminf(fmaxf(p.x * 1024.0f, 0.0f), 1023.0f);
    const float y = fminf(fmaxf(p.y * 1024.0f, 0.0f), 1023.0f);
    const float z = fminf(fmaxf(p.z * 1024.0f, 0.0f), 1023.0f);
    return expand_bits((uint32_t)x) * 4 + expand_bits((uint32_t)y) * 2 + expand_bits((uint32_t)z);
}

uint32_t lbvh_host_morton(const Box &refbox, const Box &scene)
{
    const f3 extent = scene.mx - scene.mn;
    const f3 centroid = (refbox.mn + refbox.mx) * 0.5f; // aabb.h:38-40
    const f3 rel = centroid - scene.mn;
    return morton_3d(mk3(rel.x / extent.x, rel.y / extent.y, rel.z / extent.z));
}

void reference_ranks(const std::vector<Box> &boxes, std::vector<uint32_t> &rank, Box &scene_box)
{
    const size_t n = boxes.size();
    rank.assign(n, 0);
    if (n == 0) return;
    scene_box = boxes[0]; // scene.h:167-170
    for (size_t i = 1; i < n; i++) scene_box = box_union(scene_box, boxes[i]);
    const f3 extent = scene_box.mx - scene_box.mn;
    std::vector<uint32_t> code(n), order(n);
    for (size_t i = 0; i < n; i++) { // bvh.h:35-48
        const f3 centroid = (boxes[i].mn + boxes[i].mx) * 0.5f; // aabb.h:38-40
        const f3 rel = centroid - scene_box.mn;
        code[i] = morton_3d(mk3(rel.x / extent.x, rel.y / extent.y, rel.z / extent.z));
    }
    std::iota(order.begin(), order.end(), 0u);
    // thrust::sort_by_key on uint32 keys is an LSD radix sort, hence stable (scene.h:301)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    for (size_t i = 0; i < n; i++) rank[order[i]] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------------------------
// SAH BVH
// ---------------------------------------------------------------------------------------------------------
namespace {

struct DBox {
    double mn[3], mx[3];
    void reset() { for (int k = 0; k < 3; k++) { mn[k] = 1e300; mx[k] = -1e300; } }
    void grow(const DBox &o) { for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], o.mn[k]); mx[k] = std::max(mx[k], o.mx[k]); } }
    void grow_pt(const double p[3]) { for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); } }
    double area() const
    {
        const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return (dx < 0) ? 0.0 : 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};

struct Ref { Box box; double c[3]; uint32_t tri; };

struct Builder {
    const std::vector<HostTriangle> &tris;
    const std::vector<uint32_t> &rank;
    const std::vector<BvhSphere> &spheres; // Ref.tri >= tris.size() names spheres[Ref.tri - tris.size()]
    uint32_t max_leaf;
    std::vector<Ref> refs;
    FlatBvh &out;
    static constexpr int NBINS = 64; // capacity; `nbins` are used
    static constexpr double NODE_COST = 1.0;
    double TRI_COST = 1.3;
    int nbins = 16;

    Builder(const std::vector<HostTriangle> &t, const std::vector<uint32_t> &r, const std::vector<BvhSphere> &sp, uint32_t ml, FlatBvh &o)
        : tris(t), rank(r), spheres(sp), max_leaf(ml), out(o) {}

    static DBox to_d(const Box &b)
    {
        DBox d;
        d.mn[0] = b.mn.x; d.mn[1] = b.mn.y; d.mn[2] = b.mn.z; d.mx[0] = b.mx.x; d.mx[1] = b.mx.y; d.mx[2] = b.mx.z;
        return d;
    }
    static uint32_t levels_needed(size_t n, uint32_t max_leaf)
    { // levels of nodes a median-split subtree over n triangles needs
        uint32_t l = 1;
        size_t cap = 2 * (size_t)max_leaf;
        while (cap < n) { cap *= 2; l++; }
        return l;
    }

    int make_leaf(size_t lo, size_t hi)
    {
        const uint32_t first = (uint32_t)(out.tris.size() / 3);
        for (size_t i = lo; i < hi; i++) {
            if (refs[i].tri >= tris.size()) { // sphere slot
                const BvhSphere &sp = spheres[refs[i].tri - tris.size()];
                q4 a, b, c;
                a.x = sp.s.center.x; a.y = sp.s.center.y; a.z = sp.s.center.z; b.x = sp.s.radius_arg; b.y = sp.s.inv_radius; b.z = sphere_sure_term(sp.s.center, sp.s.radius); c.x = sp.s.radius; c.y = c.z = 0.0f;
                const int id = (int)sp.id, rk = (int)(sp.rank | CRT_SPHERE_SLOT), mat = sp.s.material;
                std::memcpy(&a.w, &id, 4); std::memcpy(&b.w, &rk, 4); std::memcpy(&c.w, &mat, 4);
                out.tris.push_back(a); out.tris.push_back(b); out.tris.push_back(c);
                continue;
            }
            const HostTriangle &t = tris[refs[i].tri];
            q4 a, b, c;
            a.x = t.a.x; a.y = t.a.y; a.z = t.a.z; b.x = t.b.x; b.y = t.b.y; b.z = t.b.z; c.x = t.c.x; c.y = t.c.y; c.z = t.c.z;
            const int id = (int)refs[i].tri, rk = (int)rank[refs[i].tri], mat = t.material;
            std::memcpy(&a.w, &id, 4); std::memcpy(&b.w, &rk, 4); std::memcpy(&c.w, &mat, 4);
            out.tris.push_back(a); out.tris.push_back(b); out.tris.push_back(c);
        }
        const uint32_t count = (uint32_t)(hi - lo);
        out.max_leaf = std::max(out.max_leaf, count);
        return ~(int)((first << 3) | (count - 1));
    }

    Box range_box(size_t lo, size_t hi) const
    {
        Box b = refs[lo].box;
        for (size_t i = lo + 1; i < hi; i++) {
            const Box &o = refs[i].box;
            b.mn.x = fminf(b.mn.x, o.mn.x); b.mn.y = fminf(b.mn.y, o.mn.y); b.mn.z = fminf(b.mn.z, o.mn.z);
            b.mx.x = fmaxf(b.mx.x, o.mx.x); b.mx.y = fmaxf(b.mx.y, o.mx.y); b.mx.z = fmaxf(b.mx.z, o.mx.z);
        }
        return b;
    }

    // returns the ref of the subtree over refs[lo, hi); `levels` = node levels still available
    int build(size_t lo, size_t hi, uint32_t levels, uint32_t depth, Box &box_out)
    {
        box_out = range_box(lo, hi);
        const size_t n = hi - lo;
        if (n == 1 || (n <= max_leaf && levels == 0)) return make_leaf(lo, hi);

        DBox cb; cb.reset();
        for (size_t i = lo; i < hi; i++) cb.grow_pt(refs[i].c);
        size_t mid = 0;
        bool have_split = false;
        const bool force_median = levels <= levels_needed(n, max_leaf);
        if (!force_median) {
            double best_cost = 1e300; int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; axis++) {
                const double ext = cb.mx[axis] - cb.mn[axis];
                if (!(ext > 0)) continue;
                DBox bb[NBINS]; size_t bc[NBINS];
                for (int k = 0; k < nbins; k++) { bb[k].reset(); bc[k] = 0; }
                const double scale = nbins / ext;
                for (size_t i = lo; i < hi; i++) {
                    int k = (int)((refs[i].c[axis] - cb.mn[axis]) * scale);
                    k = std::min(std::max(k, 0), nbins - 1);
                    bb[k].grow(to_d(refs[i].box)); bc[k]++;
                }
                double right_area[NBINS]; size_t right_cnt[NBINS];
                DBox acc; acc.reset(); size_t cnt = 0;
                for (int k = nbins - 1; k > 0; k--) { acc.grow(bb[k]); cnt += bc[k]; right_area[k] = acc.area(); right_cnt[k] = cnt; }
                acc.reset(); cnt = 0;
                for (int k = 0; k < nbins - 1; k++) {
                    acc.grow(bb[k]); cnt += bc[k];
                    if (cnt == 0 || right_cnt[k + 1] == 0) continue;
                    const double cost = acc.area() * (double)cnt + right_area[k + 1] * (double)right_cnt[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = k; }
                }
            }
            if (best_axis >= 0) {
                const double parent_area = to_d(box_out).area();
                const double split_cost = NODE_COST + TRI_COST * best_cost / std::max(parent_area, 1e-300);
                const double leaf_cost = TRI_COST * (double)n;
                if (n <= max_leaf && leaf_cost <= split_cost) return make_leaf(lo, hi);
                const double ext = cb.mx[best_axis] - cb.mn[best_axis];
                const double scale = nbins / ext;
                const double mn = cb.mn[best_axis];
                auto it = std::partition(refs.begin() + (long)lo, refs.begin() + (long)hi, [&](const Ref &r) {
                    int k = (int)((r.c[best_axis] - mn) * scale);
                    k = std::min(std::max(k, 0), nbins - 1);
                    return k <= best_bin;
                });
                mid = (size_t)(it - refs.begin());
                have_split = mid > lo && mid < hi;
            } else if (n <= max_leaf) {
                return make_leaf(lo, hi); // all centroids coincide
            }
        }
        if (!have_split) { // object median along the widest centroid axis (also the depth-bounded fallback)
            int axis = 0;
            double w = -1;
            for (int k = 0; k < 3; k++) if (cb.mx[k] - cb.mn[k] > w) { w = cb.mx[k] - cb.mn[k]; axis = k; }
            mid = lo + n / 2;
            std::nth_element(refs.begin() + (long)lo, refs.begin() + (long)mid, refs.begin() + (long)hi,
                             [&](const Ref &a, const Ref &b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.tri < b.tri); });
        }
        const int node = (int)(out.nodes.size() / 4);
        out.nodes.resize(out.nodes.size() + 4);
        out.depth = std::max(out.depth, depth);
        Box b0, b1;
        const uint32_t next_levels = levels > 0 ? levels - 1 : 0;
        const int r0 = build(lo, mid, next_levels, depth + 1, b0);
        const int r1 = build(mid, hi, next_levels, depth + 1, b1);
        q4 *q = &out.nodes[4 * (size_t)node];
        q[0].x = b0.mn.x; q[0].y = b0.mn.y; q[0].z = b0.mn.z; std::memcpy(&q[0].w, &r0, 4);
        q[1].x = b0.mx.x; q[1].y = b0.mx.y; q[1].z = b0.mx.z; std::memcpy(&q[1].w, &r1, 4);
        q[2].x = b1.mn.x; q[2].y = b1.mn.y; q[2].z = b1.mn.z; q[2].w = 0.0f;
        q[3].x = b1.mx.x; q[3].y = b1.mx.y; q[3].z = b1.mx.z; q[3].w = 0.0f;
        return node;
    }
};

} // namespace

// Breadth-first node numbering: node index < N  <=>  one of the N nodes nearest the root, so "the top of the tree" is a
// prefix of the array (the kernels keep that prefix in LDS).  Records that no reference reaches (the GPU builder leaves
// the nodes of collapsed sub-trees behind) are dropped.  Triangle slots keep their order.
void renumber_breadth_first(std::vector<q4> &nodes_io, int &root_ref)
{
    if (root_ref < 0 || root_ref == CRT_BVH_NONE) return;
    const size_t n = nodes_io.size() / 4;
    std::vector<int> order;        // new index -> old index
    std::vector<int> renum(n, -1); // old index -> new index
    order.reserve(n);
    order.push_back(root_ref);
    renum[(size_t)root_ref] = 0;
    for (size_t head = 0; head < order.size(); head++) {
        const q4 *q = &nodes_io[4 * (size_t)order[head]];
        int r[2];
        std::memcpy(&r[0], &q[0].w, 4); std::memcpy(&r[1], &q[1].w, 4);
        for (int k = 0; k < 2; k++)
            if (r[k] >= 0) { renum[(size_t)r[k]] = (int)order.size(); order.push_back(r[k]); }
    }
    std::vector<q4> nodes(4 * order.size());
    for (size_t k = 0; k < order.size(); k++) {
        const q4 *src = &nodes_io[4 * (size_t)order[k]];
        q4 *dst = &nodes[4 * k];
        for (int j = 0; j < 4; j++) dst[j] = src[j];
        int r[2];
        std::memcpy(&r[0], &src[0].w, 4); std::memcpy(&r[1], &src[1].w, 4);
        for (int j = 0; j < 2; j++) if (r[j] >= 0) r[j] = renum[(size_t)r[j]];
        std::memcpy(&dst[0].w, &r[0], 4); std::memcpy(&dst[1].w, &r[1], 4);
    }
    nodes_io.swap(nodes);
    root_ref = 0;
}

float bvh_sah_cost(const FlatBvh &bvh)
{
    if (bvh.wide || bvh.root_ref < 0 || bvh.root_ref == CRT_BVH_NONE) return 0.0f;
    auto area = [](const q4 &mn, const q4 &mx) {
        const double dx = (double)mx.x - mn.x, dy = (double)mx.y - mn.y, dz = (double)mx.z - mn.z;
        return dx * dy + dy * dz + dz * dx;
    };
    q4 rmn, rmx;
    rmn.x = bvh.root_box.mn.x; rmn.y = bvh.root_box.mn.y; rmn.z = bvh.root_box.mn.z; rmx.x = bvh.root_box.mx.x; rmx.y = bvh.root_box.mx.y; rmx.z = bvh.root_box.mx.z;
    rmn.w = rmx.w = 0.0f;
    const double root_area = area(rmn, rmx);
    if (!(root_area > 0)) return 0.0f;
    double cost = root_area; // the root's own visit
    const size_t n = bvh.nodes.size() / 4;
    for (size_t i = 0; i < n; i++) {
        const q4 *q = &bvh.nodes[4 * i];
        for (int k = 0; k < 2; k++) {
            int ref;
            std::memcpy(&ref, &q[k].w, 4);
            const double a = area(q[2 * k], q[2 * k + 1]);
            if (ref >= 0) cost += a;
            else if (ref != CRT_BVH_NONE) cost += 1.3 * (double)(((~ref) & 7) + 1) * a;
        }
    }
    return (float)(cost / root_area);
}

void build_bvh(const std::vector<HostTriangle> &tris, const std::vector<uint32_t> &rank, const std::vector<BvhSphere> &spheres, uint32_t max_leaf,
               FlatBvh &out)
{
    out = FlatBvh();
    if (tris.empty() && spheres.empty()) return;
    max_leaf = std::min(std::max(max_leaf, 1u), 8u);
    // Rounding margin of the pruning slab test (crt_traverse.h): the ray origin is re-based to the root box, so
    // plane times carry an absolute error of a few ulp of the largest coordinate; 2^-18 * max|coord| covers it
    // with a factor > 4 to spare, on top of the reference's own 1e-4 padding.
    float maxabs = 0.0f;
    for (const HostTriangle &t : tris) {
        const float v[9] = { t.a.x, t.a.y, t.a.z, t.b.x, t.b.y, t.b.z, t.c.x, t.c.y, t.c.z };
        for (float f : v) maxabs = fmaxf(maxabs, fabsf(f));
    }
    for (const BvhSphere &sp : spheres) {
        const float r = fmaxf(sp.s.radius, fabsf(sp.s.radius_arg));
        maxabs = fmaxf(maxabs, fmaxf(fmaxf(fabsf(sp.s.center.x), fabsf(sp.s.center.y)), fabsf(sp.s.center.z)) + r);
    }
    const float margin = fmaxf(1e-5f, maxabs * 3.814697265625e-06f);
    Builder b(tris, rank, spheres, max_leaf, out);
    if (const char *e = getenv("CRT_SAH_TRI_COST")) b.TRI_COST = atof(e);
    if (const char *e = getenv("CRT_SAH_BINS")) b.nbins = std::min(std::max(atoi(e), 2), 64);
    b.refs.resize(tris.size() + spheres.size());
    for (size_t i = 0; i < b.refs.size(); i++) {
        Box bx;
        if (i < tris.size()) bx = reference_triangle_box(tris[i]);
        else { // every point of the sphere (radius = max(0, argument), sphere.h:11) and the reference's own box
            const HostSphere &sp = spheres[i - tris.size()].s;
            const float r = fmaxf(sp.radius, fabsf(sp.radius_arg));
            bx.mn = sp.center - mk3(r, r, r); bx.mx = sp.center + mk3(r, r, r);
        }
        bx.mn = bx.mn - mk3(margin, margin, margin);
        bx.mx = bx.mx + mk3(margin, margin, margin);
        b.refs[i].box = bx;
        b.refs[i].c[0] = 0.5 * ((double)bx.mn.x + bx.mx.x); b.refs[i].c[1] = 0.5 * ((double)bx.mn.y + bx.mx.y);
        b.refs[i].c[2] = 0.5 * ((double)bx.mn.z + bx.mx.z);
        b.refs[i].tri = (uint32_t)i;
    }
    out.nodes.reserve(4 * b.refs.size());
    out.tris.reserve(3 * b.refs.size());
    out.root_ref = b.build(0, b.refs.size(), CRT_STACK_DEPTH, 1, out.root_box);

    renumber_breadth_first(out.nodes, out.root_ref);
}

// Two-child tree -> four-child tree (crt_traverse.h: trav_inner4_from).  A wide node starts as a two-child node's pair of
// children; the inner child with the largest box is replaced by its own two children until there are four or only leaves are
// left (the usual collapse: every root-to-leaf path loses about half of its levels).  Nodes are emitted breadth first, so "the
// first N nodes" is still the top of the tree; a reference to a wide node carries the mask of the children it has.
void widen_bvh(FlatBvh &bvh)
{
    if (bvh.wide || bvh.root_ref < 0 || bvh.root_ref == CRT_BVH_NONE) return; // (no node at all: a single leaf, or nothing)
    struct Child { Box box; int ref; };
    auto child_of = [&](int node, int k) {
        const q4 *q = &bvh.nodes[4 * (size_t)node];
        Child c;
        c.box.mn = mk3(q[2 * k].x, q[2 * k].y, q[2 * k].z); c.box.mx = mk3(q[2 * k + 1].x, q[2 * k + 1].y, q[2 * k + 1].z);
        std::memcpy(&c.ref, &q[k].w, 4);
        return c;
    };
    auto area = [](const Box &b) {
        const double dx = (double)b.mx.x - b.mn.x, dy = (double)b.mx.y - b.mn.y, dz = (double)b.mx.z - b.mn.z;
        return dx * dy + dy * dz + dz * dx;
    };
    struct Wide { Child c[4]; int n; };
    std::vector<Wide> wide;            // breadth first
    std::vector<int> source;           // two-child node a wide node was grown from
    std::vector<int> wide_of(bvh.nodes.size() / 4, -1);
    source.push_back(bvh.root_ref); wide_of[(size_t)bvh.root_ref] = 0;
    uint32_t depth = 0;
    for (size_t head = 0; head < source.size(); head++) {
        Wide w;
        w.c[0] = child_of(source[head], 0); w.c[1] = child_of(source[head], 1); w.n = 2;
        while (w.n < 4) {
            int best = -1;
            double best_area = -1.0;
            for (int k = 0; k < w.n; k++)
                if (w.c[k].ref >= 0 && area(w.c[k].box) > best_area) { best_area = area(w.c[k].box); best = k; }
            if (best < 0) break;
            const int node = w.c[best].ref;
            w.c[best] = child_of(node, 0);
            w.c[w.n++] = child_of(node, 1);
        }
        wide.push_back(w);
        for (int k = 0; k < w.n; k++)
            if (w.c[k].ref >= 0) { wide_of[(size_t)w.c[k].ref] = (int)source.size(); source.push_back(w.c[k].ref); }
    }
    if (wide.size() > (size_t)CRT_WIDE_INDEX) return; // (more nodes than a reference can name: stays two-child)
    std::vector<uint32_t> level(wide.size(), 1); // parents come before their children
    std::vector<q4> out(8 * wide.size());
    for (size_t i = 0; i < wide.size(); i++) {
        const Wide &w = wide[i];
        depth = std::max(depth, level[i]);
        float f[7][4];
        for (int k = 0; k < 4; k++) {
            const bool have = k < w.n;
            const Box b = have ? w.c[k].box : Box{ mk3(0, 0, 0), mk3(0, 0, 0) }; // (a child that does not exist is never named by a mask)
            f[0][k] = b.mn.x; f[1][k] = b.mn.y; f[2][k] = b.mn.z; f[3][k] = b.mx.x; f[4][k] = b.mx.y; f[5][k] = b.mx.z;
            int ref = CRT_BVH_NONE;
            if (have) {
                ref = w.c[k].ref;
                if (ref >= 0) {
                    const int child = wide_of[(size_t)ref];
                    level[(size_t)child] = level[i] + 1;
                    ref = (int)(((uint32_t)((1 << wide[(size_t)child].n) - 1) << CRT_WIDE_SHIFT) | (uint32_t)child);
                }
            }
            std::memcpy(&f[6][k], &ref, 4);
        }
        for (int r = 0; r < 7; r++) { q4 &q = out[8 * i + (size_t)r]; q.x = f[r][0]; q.y = f[r][1]; q.z = f[r][2]; q.w = f[r][3]; }
        out[8 * i + 7] = q4{ 0, 0, 0, 0 };
    }
    bvh.nodes.swap(out);
    bvh.root_ref = (int)(((uint32_t)((1 << wide[0].n) - 1) << CRT_WIDE_SHIFT) | 0u);
    bvh.depth = depth;
    bvh.wide = true;
}

float reciprocal_if_exact(float s)
{
    if (!(s >= 0x1p-20f && s <= 0x1p+20f)) return 0.0f;
    static std::mutex mutex;
    static std::map<uint32_t, float> known;
    uint32_t key;
    std::memcpy(&key, &s, 4);
    {
        std::lock_guard<std::mutex> lock(mutex);
        auto it = known.find(key);
        if (it != known.end()) return it->second;
        if (known.size() >= 256) return 0.0f; // a scene with that many different radii: the rest divide the ordinary way
    }
    const float y = 1.0f / s;
    // every significand at one exponent (both signs behave alike), shared among a few threads: 2^23 cases take 65 ms on one core,
    // and a scene's first commit checks each of its radii
    const unsigned hw = std::thread::hardware_concurrency();
    const uint32_t nthreads = std::min(std::max(hw, 1u), 8u), total = 1u << 23, per = total / nthreads;
    std::atomic<bool> exact{ true };
    auto check = [&](uint32_t m0, uint32_t m1) {
        for (uint32_t m = m0; m < m1; m++) {
            if ((m & 0xFFFFu) == 0 && !exact.load(std::memory_order_relaxed)) return; // (someone found a counter-example)
            const uint32_t bits = 0x40000000u | m;
            float a;
            std::memcpy(&a, &bits, 4);
            const float q = a * y, got = __builtin_fmaf(__builtin_fmaf(-s, q, a), y, q), want = a / s;
            if (std::memcmp(&got, &want, 4) != 0) { exact.store(false, std::memory_order_relaxed); return; }
        }
    };
    {
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < nthreads; t++) pool.emplace_back(check, t * per, t + 1 == nthreads ? total : (t + 1) * per);
        check(0, nthreads == 1 ? total : per);
        for (std::thread &t : pool) t.join();
    }
    const float result = exact ? y : 0.0f;
    std::lock_guard<std::mutex> lock(mutex);
    known[key] = result;
    return result;
}

} // namespace crt
