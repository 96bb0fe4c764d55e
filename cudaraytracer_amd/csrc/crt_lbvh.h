// crt_lbvh.h -- GPU construction of ranks + triangle BVH (crt_lbvh.hip); called by crt_scene_commit when asked for.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <vector>

#include "crt_bvh.h"

namespace crt {

struct LbvhSphereCodes { uint32_t count; uint32_t code[8]; }; // Morton codes of the scene's (few) spheres
struct LbvhSphereInput { Box refbox; };                       // reference box of a sphere (sphere.h:12-15)

struct LbvhOutput {
    std::vector<q4> nodes;              // 4 per node, Karras numbering / allocation order (records of collapsed sub-trees are zero)
    std::vector<q4> slots;              // 3 per triangle, in sorted (rank) order
    std::vector<q4> normals;            // per slot (SAH tree only; else empty: the host computes them)
    bool breadth_first = false;         // nodes are numbered breadth first already (SAH tree)
    // in: keep the slots and normals of a SAH tree on the device instead of bringing them to the host (`slots` / `normals` stay
    // empty); out: the two buffers, the caller's to hipFree (or to render from: they are what a replica on that device holds)
    bool keep_on_device = false;
    void *d_slots = nullptr, *d_normals = nullptr;
    std::vector<uint32_t> rank_of_tri;  // global rank of triangle i
    std::vector<uint32_t> sorted_codes; // the triangles' codes in sorted order (for the spheres' ranks)
    LbvhSphereCodes sphere_codes{};
    Box scene_box{}, root_box{};
    int root_ref = CRT_BVH_NONE;
    uint32_t depth = 0;                 // levels of nodes on the longest path
    float build_ms = 0.0f;              // hipEvent time of the build kernels + sort (uploads / downloads excluded)
};

// bvh.h:24-48 on the host, for the spheres (defined in crt_bvh.cpp next to the code it shares with reference_ranks)
uint32_t lbvh_host_morton(const Box &refbox, const Box &scene);

// The tree over the sorted triangles: Karras LBVH (crt_lbvh.hip) or binned SAH by levels (crt_gsah.hip)
enum { LBVH_TREE_KARRAS = 0, LBVH_TREE_SAH = 1 };

// crt_gsah.hip: binned-SAH tree over n triangles on the device.  d_refbox6: the reference boxes (6 floats each, min xyz max
// xyz); d_order: a starting order of the triangles (any permutation; Morton order makes the rare median splits spatial);
// tri_box: union of the reference boxes.  d_records: scratch for n + 1 records (allocation order; record n receives the root's
// box and reference); d_nodes_out: the res.num_records nodes in breadth-first numbering (room for n); d_slots / d_normals: the
// 48-byte slots and their unit normals (crt_traverse.h: triangle_normal) in leaf order.
struct GsahResult { uint32_t num_records = 0, depth = 0, levels_run = 0, small_nodes = 0; int root_ref = CRT_BVH_NONE; Box root_box{}; };
hipError_t gsah_build(const float *d_tri9, const int *d_mat, const float *d_refbox6, const uint32_t *d_order, const uint32_t *d_rank_of_tri, uint32_t n,
                      float margin, const float tri_box[6], uint32_t max_leaf, float tri_cost, q4 *d_records, q4 *d_nodes_out, q4 *d_slots, q4 *d_normals, GsahResult &res,
                      hipStream_t stream);

// h_tri9: 9 floats per triangle (a, b, c); h_mat: material per triangle.  sphere_union: union of the spheres' reference
// boxes (have_spheres false: none).  At most 8 spheres.  The device is the current one.
hipError_t lbvh_build(const float *h_tri9, const int *h_mat, uint32_t n, const Box &sphere_union, bool have_spheres, const LbvhSphereInput *spheres,
                      uint32_t nspheres, uint32_t max_leaf, uint32_t tree_kind, LbvhOutput &out, hipStream_t stream);

} // namespace crt
