// crt_bvh.h -- host-side scene preparation: reference primitive boxes / Morton ranks (the part of the
// reference's setup that decides hot-path results) and this library's own SAH BVH, flattened to 64-byte
// two-child nodes and 48-byte triangle slots.
#pragma once

#include <stdint.h>
#include <vector>

#include "crt_traverse.h"

namespace crt {

struct HostTriangle { f3 a, b, c; int material; };
struct HostSphere { f3 center; float radius_arg; float radius; int material; float inv_radius = 0.0f; }; // radius = max(0, radius_arg) (sphere.h:11); inv_radius: reciprocal_if_exact(radius), set at commit

// RN(1 / s) if crt_core.h's div3_exact with it equals IEEE division for every numerator significand (checked exhaustively,
// 2^23 cases, ~10 ms; results are kept per divisor), else 0.  Divisors outside [2^-20, 2^20] get 0.
float reciprocal_if_exact(float s);

struct Box { f3 mn, mx; };

// Triangle::init (triangle.h:28-46): min/max of the vertices -/+ 1e-4
Box reference_triangle_box(const HostTriangle &t);
// Sphere ctor (sphere.h:12-15): AABB(center - rvec, center + rvec) with the *unclamped* radius argument
Box reference_sphere_box(const HostSphere &s);

// rank[i] = position of primitive i (triangles, then spheres) in the reference's stable sort by 30-bit Morton
// code of the box centroid normalised to the scene box (scene.h:167-170, bvh.h:13-48, scene.h:299-301).
// This is the DFS leaf order of the reference LBVH (bvh.h:91-159), i.e. the order that decides ties.
void reference_ranks(const std::vector<Box> &boxes, std::vector<uint32_t> &rank, Box &scene_box);

struct FlatBvh {
    std::vector<q4> nodes;   // 4 per two-child node, 8 per four-child node
    bool wide = false;       // four-child nodes (widen_bvh)
    std::vector<q4> tris;    // 3 per slot
    int root_ref = CRT_BVH_NONE;
    Box root_box{};
    uint32_t depth = 0;      // levels of nodes on the longest root-to-leaf path
    uint32_t max_leaf = 0;
};

// A sphere that lives in the BVH (scenes with many spheres, scene.h:53-90): its primitive id and reference rank.
struct BvhSphere { HostSphere s; uint32_t id; uint32_t rank; };

// SAH (binned) BVH over the primitives' reference boxes, inflated by a rounding margin.  rank/ids are carried
// into the 48-byte slots; a sphere slot is [centre, id][radius argument, inv_radius, sphere_sure_term, rank | CRT_SPHERE_SLOT][radius, 0, 0, material].
// rank[i] is the rank of triangle i.  max_leaf in 1..8.
void build_bvh(const std::vector<HostTriangle> &tris, const std::vector<uint32_t> &rank, const std::vector<BvhSphere> &spheres, uint32_t max_leaf,
               FlatBvh &out);

// Two-child tree -> four-child tree of 128-byte nodes (crt_traverse.h), breadth first; depth becomes the wide tree's.
void widen_bvh(FlatBvh &bvh);

// Surface-area cost of a two-child tree: 1 per node visited + 1.3 per triangle tested, each weighted by the area of its box
// over the root box's (the builders' own objective; lower = fewer expected steps per ray).  0 for a tree without nodes.
float bvh_sah_cost(const FlatBvh &bvh);

// Breadth-first renumbering of a flattened tree (drops unreachable records): what "the first N nodes" means to the kernels.
void renumber_breadth_first(std::vector<q4> &nodes, int &root_ref);

} // namespace crt
