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
    std::vector<q4> nodes;              // 4 per node, Karras numbering (records of collapsed sub-trees are zero)
    std::vector<q4> slots;              // 3 per triangle, in sorted (rank) order
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

// h_tri9: 9 floats per triangle (a, b, c); h_mat: material per triangle.  sphere_union: union of the spheres' reference
// boxes (have_spheres false: none).  At most 8 spheres.  The device is the current one.
hipError_t lbvh_build(const float *h_tri9, const int *h_mat, uint32_t n, const Box &sphere_union, bool have_spheres, const LbvhSphereInput *spheres,
                      uint32_t nspheres, uint32_t max_leaf, LbvhOutput &out, hipStream_t stream);

} // namespace crt
