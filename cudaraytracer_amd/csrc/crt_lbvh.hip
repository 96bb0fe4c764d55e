// crt_lbvh.hip -- GPU construction of the triangle BVH (SURVEY section 8 "next" row N1; reference: bvh.h:13-185 and
// scene.h:288-307, a Morton sort followed by a single-thread build_bvh<<<1,1>>>).
//
// What it produces is the same pair the host path produces (crt_bvh.cpp): the reference's *ranks* (position of every
// primitive in the stable sort by the reference's 30-bit Morton code: this is what decides ties) and a tree in this
// library's 64-byte two-child node / 48-byte slot format.  Here the tree is the LBVH over that very order (Karras 2012:
// one thread per internal node finds its range and split from the sorted codes; boxes are fitted bottom-up with one
// atomic flag per node), with sub-trees of at most `max_leaf` triangles collapsed into leaves.  It is built in a few
// milliseconds instead of seconds, and traverses slower than the host's SAH tree; results are identical either way
// because boxes only prune (crt_traverse.h).
//
// Exactness of the ranks: boxes, centroids, normalisation and the Morton code are evaluated with the reference's float
// operations in the reference's order (no contraction, IEEE division), as crt_bvh.cpp does on the host;
// rocprim::radix_sort_pairs is a stable LSD radix sort like thrust::sort_by_key (scene.h:301).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "crt_bvh.h"
#include "crt_lbvh.h"

namespace crt {

namespace {

#define LBVH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return e_; } } while (0)

// order-preserving float <-> uint mapping for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : u | 0x80000000u; }
inline float ord2f(uint32_t o) { const uint32_t u = (o & 0x80000000u) ? o & 0x7FFFFFFFu : ~o; float f; memcpy(&f, &u, 4); return f; }

__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(__builtin_fminf(a, a), b), c); } // utility.h:45-51
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a, a), b), c); } // utility.h:53-59

struct Box6 { float mnx, mny, mnz, mxx, mxy, mxz; };

// Triangle::init (triangle.h:28-46) through AABB(point, point) (aabb.h:12-16): the reference box of triangle i,
// and the wave/block reduction of the scene box and of the largest |coordinate|.  red: [0..2] min, [3..5] max, [6] maxabs
__global__ __launch_bounds__(256) void tri_boxes_kernel(const float *__restrict__ tri9, uint32_t n, Box6 *__restrict__ refbox, uint32_t *__restrict__ red)
{
    // (a bounded grid that strides over the triangles: the seven atomics per wave all go to the same seven words, and a million
    // triangles' worth of them took 1.2 ms; 2048 work-groups reduce in registers first)
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX }, maxabs = 0.0f;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float *t = tri9 + 9 * (size_t)i;
        const float EPSILON = 1e-4f;
        Box6 b;
        float p[3], q[3];
        for (int k = 0; k < 3; k++) {
            p[k] = min3f(t[k], t[3 + k], t[6 + k]) - EPSILON;
            q[k] = max3f(t[k], t[3 + k], t[6 + k]) + EPSILON;
            maxabs = __builtin_fmaxf(maxabs, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(t[k]), __builtin_fabsf(t[3 + k])), __builtin_fabsf(t[6 + k])));
        }
        // aabb.h:12-16: per axis the smaller coordinate becomes the interval minimum
        b.mnx = p[0] < q[0] ? p[0] : q[0]; b.mxx = p[0] < q[0] ? q[0] : p[0];
        b.mny = p[1] < q[1] ? p[1] : q[1]; b.mxy = p[1] < q[1] ? q[1] : p[1];
        b.mnz = p[2] < q[2] ? p[2] : q[2]; b.mxz = p[2] < q[2] ? q[2] : p[2];
        refbox[i] = b;
        mn[0] = __builtin_fminf(mn[0], b.mnx); mn[1] = __builtin_fminf(mn[1], b.mny); mn[2] = __builtin_fminf(mn[2], b.mnz);
        mx[0] = __builtin_fmaxf(mx[0], b.mxx); mx[1] = __builtin_fmaxf(mx[1], b.mxy); mx[2] = __builtin_fmaxf(mx[2], b.mxz);
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int k = 0; k < 3; k++) {
            mn[k] = __builtin_fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = __builtin_fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
        maxabs = __builtin_fmaxf(maxabs, __shfl_xor(maxabs, off, 64));
    }
    if ((threadIdx.x & 63u) == 0 && mn[0] <= mx[0]) { // (a wave that saw no triangle has nothing to say)
        for (int k = 0; k < 3; k++) { atomicMin(&red[k], f2ord(mn[k])); atomicMax(&red[3 + k], f2ord(mx[k])); }
        atomicMax(&red[6], __float_as_uint(maxabs));
    }
}

// bvh.h:13-20
__device__ __forceinline__ uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// bvh.h:24-48: Morton code of the box centroid normalised to the scene box
__global__ __launch_bounds__(256) void tri_codes_kernel(const Box6 *__restrict__ refbox, uint32_t n, float smx, float smy, float smz, float ex, float ey, float ez,
                                                        uint32_t *__restrict__ code, uint32_t *__restrict__ idx)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const Box6 b = refbox[i];
    const float cx = (b.mnx + b.mxx) * 0.5f, cy = (b.mny + b.mxy) * 0.5f, cz = (b.mnz + b.mxz) * 0.5f; // aabb.h:38-40
    const float px = (cx - smx) / ex, py = (cy - smy) / ey, pz = (cz - smz) / ez;
    const float x = __builtin_fminf(__builtin_fmaxf(px * 1024.0f, 0.0f), 1023.0f);
    const float y = __builtin_fminf(__builtin_fmaxf(py * 1024.0f, 0.0f), 1023.0f);
    const float z = __builtin_fminf(__builtin_fmaxf(pz * 1024.0f, 0.0f), 1023.0f);
    code[i] = expand_bits((uint32_t)x) * 4 + expand_bits((uint32_t)y) * 2 + expand_bits((uint32_t)z);
    idx[i] = i;
}

// global rank of every triangle from the reference-order sort: triangles precede spheres in the reference's list
// (scene.h:161-165) and the sort is stable, so only the spheres with a strictly smaller code come before a triangle
__global__ __launch_bounds__(256) void ranks_kernel(const uint32_t *__restrict__ code_s, const uint32_t *__restrict__ idx_s, uint32_t n, LbvhSphereCodes sph,
                                                    uint32_t *__restrict__ rank_of_tri)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    uint32_t rank = p;
    const uint32_t c = code_s[p];
    for (uint32_t k = 0; k < sph.count; k++) rank += sph.code[k] < c ? 1u : 0u;
    rank_of_tri[idx_s[p]] = rank;
}

// The tree's own order: 30-bit Morton code of the centroid normalised to the box of the TRIANGLES.  (The reference
// normalises to the box of everything, scene.h:167-170; with its r = 1000 ground sphere the mesh falls into a few
// cells of the 1024^3 grid, F10, which is fine for ranks -- they only break ties -- and poor for a tree.)
__device__ __forceinline__ unsigned long long expand_bits_21(unsigned long long v) // 21 bits -> every third bit
{
    v &= 0x1FFFFFull;
    v = (v | v << 32) & 0x1F00000000FFFFull;
    v = (v | v << 16) & 0x1F0000FF0000FFull;
    v = (v | v << 8) & 0x100F00F00F00F00Full;
    v = (v | v << 4) & 0x10C30C30C30C30C3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
__global__ __launch_bounds__(256) void build_codes_kernel(const Box6 *__restrict__ refbox, uint32_t n, float tmx, float tmy, float tmz, float sx, float sy, float sz,
                                                          unsigned long long *__restrict__ code, uint32_t *__restrict__ idx)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const Box6 b = refbox[i];
    // 63-bit codes (2^21 cells per axis): with 10 bits per axis a million triangles share cells and the tie-break by
    // position builds poor sub-trees (measured: 162 ms instead of 60 ms per frame on the 1M-triangle grid)
    const float x = __builtin_fminf(__builtin_fmaxf(((b.mnx + b.mxx) * 0.5f - tmx) * sx, 0.0f), 2097151.0f);
    const float y = __builtin_fminf(__builtin_fmaxf(((b.mny + b.mxy) * 0.5f - tmy) * sy, 0.0f), 2097151.0f);
    const float z = __builtin_fminf(__builtin_fmaxf(((b.mnz + b.mxz) * 0.5f - tmz) * sz, 0.0f), 2097151.0f);
    code[i] = expand_bits_21((unsigned long long)x) << 2 | expand_bits_21((unsigned long long)y) << 1 | expand_bits_21((unsigned long long)z);
    idx[i] = i;
}

// common-prefix length of sorted positions i and j (position breaks ties between equal codes); -1 outside the array
__device__ __forceinline__ int delta(const unsigned long long *__restrict__ code, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = code[i], b = code[j];
    return a == b ? 64 + __clz((uint32_t)i ^ (uint32_t)j) : __clzll(a ^ b);
}

#define LBVH_LEAF 0x80000000u // child is a sorted position, not an internal node

// Karras 2012, one thread per internal node i in [0, n-2]: its range [first, last] of sorted positions and its two children
__global__ __launch_bounds__(256) void hierarchy_kernel(const unsigned long long *__restrict__ code, int n, uint32_t *__restrict__ child0, uint32_t *__restrict__ child1,
                                                        uint32_t *__restrict__ first, uint32_t *__restrict__ last, uint32_t *__restrict__ parent_of_node,
                                                        uint32_t *__restrict__ parent_of_leaf)
{
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    const int d = delta(code, n, i, i + 1) - delta(code, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(code, n, i, i - d);
    int lmax = 2;
    while (delta(code, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(code, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(code, n, i, j);
    int s = 0, t = l;
    do { // t = ceil(l/2), ceil(l/4), ..., 1
        t = (t + 1) >> 1;
        if (delta(code, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t c0 = lo == gamma ? (uint32_t)gamma | LBVH_LEAF : (uint32_t)gamma;
    const uint32_t c1 = hi == gamma + 1 ? (uint32_t)(gamma + 1) | LBVH_LEAF : (uint32_t)(gamma + 1);
    child0[i] = c0; child1[i] = c1; first[i] = (uint32_t)lo; last[i] = (uint32_t)hi;
    if (c0 & LBVH_LEAF) parent_of_leaf[gamma] = (uint32_t)i; else parent_of_node[gamma] = (uint32_t)i;
    if (c1 & LBVH_LEAF) parent_of_leaf[gamma + 1] = (uint32_t)i; else parent_of_node[gamma + 1] = (uint32_t)i;
    if (i == 0) parent_of_node[0] = 0xFFFFFFFFu;
}

__device__ __forceinline__ Box6 box_union(const Box6 &a, const Box6 &b)
{
    Box6 r;
    r.mnx = __builtin_fminf(a.mnx, b.mnx); r.mny = __builtin_fminf(a.mny, b.mny); r.mnz = __builtin_fminf(a.mnz, b.mnz);
    r.mxx = __builtin_fmaxf(a.mxx, b.mxx); r.mxy = __builtin_fmaxf(a.mxy, b.mxy); r.mxz = __builtin_fmaxf(a.mxz, b.mxz);
    return r;
}

// Leaf boxes (reference box + rounding margin, as crt_bvh.cpp) and the bottom-up fit: the second thread to reach a node
// owns it.  Also writes the 48-byte slots in sorted order.
__global__ __launch_bounds__(256) void fit_kernel(const float *__restrict__ tri9, const int *__restrict__ mat, const Box6 *__restrict__ refbox,
                                                  const uint32_t *__restrict__ idx, int n, float margin, const uint32_t *__restrict__ child0, const uint32_t *__restrict__ child1,
                                                  const uint32_t *__restrict__ parent_of_node, const uint32_t *__restrict__ parent_of_leaf,
                                                  Box6 *__restrict__ leafbox, Box6 *__restrict__ nodebox, uint32_t *__restrict__ flags,
                                                  q4 *__restrict__ slots, const uint32_t *__restrict__ rank_of_tri)
{
    const int p = (int)(blockIdx.x * 256u + threadIdx.x);
    if (p >= n) return;
    const uint32_t t = idx[p];
    Box6 b = refbox[t];
    b.mnx -= margin; b.mny -= margin; b.mnz -= margin; b.mxx += margin; b.mxy += margin; b.mxz += margin;
    leafbox[p] = b;
    const uint32_t rank = rank_of_tri[t];
    const float *v = tri9 + 9 * (size_t)t;
    q4 a, bb, cc;
    a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = __int_as_float((int)t);
    bb.x = v[3]; bb.y = v[4]; bb.z = v[5]; bb.w = __int_as_float((int)rank);
    cc.x = v[6]; cc.y = v[7]; cc.z = v[8]; cc.w = __int_as_float(mat[t]);
    slots[3 * (size_t)p + 0] = a; slots[3 * (size_t)p + 1] = bb; slots[3 * (size_t)p + 2] = cc;
    if (n == 1) return;
    __threadfence();
    uint32_t node = parent_of_leaf[p];
    while (node != 0xFFFFFFFFu) {
        if (atomicAdd(&flags[node], 1u) == 0u) return; // first to arrive: the sibling's thread will do this node
        __threadfence();
        const uint32_t c0 = child0[node], c1 = child1[node];
        const Box6 b0 = (c0 & LBVH_LEAF) ? leafbox[c0 & ~LBVH_LEAF] : nodebox[c0];
        const Box6 b1 = (c1 & LBVH_LEAF) ? leafbox[c1 & ~LBVH_LEAF] : nodebox[c1];
        nodebox[node] = box_union(b0, b1);
        __threadfence();
        node = parent_of_node[node];
    }
}

// Node records: sub-trees of at most max_leaf triangles are leaves.  Node numbering = Karras numbering (unreachable
// records of collapsed sub-trees stay zero).
__global__ __launch_bounds__(256) void emit_kernel(int n, uint32_t max_leaf, const uint32_t *__restrict__ child0, const uint32_t *__restrict__ child1,
                                                   const uint32_t *__restrict__ first, const uint32_t *__restrict__ last, const Box6 *__restrict__ leafbox,
                                                   const Box6 *__restrict__ nodebox, q4 *__restrict__ nodes)
{
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    q4 r[4];
    for (int k = 0; k < 4; k++) { r[k].x = r[k].y = r[k].z = r[k].w = 0.0f; }
    if (last[i] - first[i] + 1u > max_leaf) {
        const uint32_t c[2] = { child0[i], child1[i] };
        int ref[2];
        Box6 b[2];
        for (int k = 0; k < 2; k++) {
            if (c[k] & LBVH_LEAF) {
                const uint32_t p = c[k] & ~LBVH_LEAF;
                b[k] = leafbox[p];
                ref[k] = ~(int)((p << 3) | 0u);
            } else {
                b[k] = nodebox[c[k]];
                const uint32_t cnt = last[c[k]] - first[c[k]] + 1u;
                ref[k] = cnt <= max_leaf ? ~(int)((first[c[k]] << 3) | (cnt - 1u)) : (int)c[k];
            }
        }
        r[0].x = b[0].mnx; r[0].y = b[0].mny; r[0].z = b[0].mnz; r[0].w = __int_as_float(ref[0]);
        r[1].x = b[0].mxx; r[1].y = b[0].mxy; r[1].z = b[0].mxz; r[1].w = __int_as_float(ref[1]);
        r[2].x = b[1].mnx; r[2].y = b[1].mny; r[2].z = b[1].mnz;
        r[3].x = b[1].mxx; r[3].y = b[1].mxy; r[3].z = b[1].mxz;
    }
    for (int k = 0; k < 4; k++) nodes[4 * (size_t)i + k] = r[k];
}

// levels of (uncollapsed) nodes above each sorted position: bounds the traversal stack
__global__ __launch_bounds__(256) void depth_kernel(int n, uint32_t max_leaf, const uint32_t *__restrict__ first, const uint32_t *__restrict__ last,
                                                    const uint32_t *__restrict__ parent_of_node, const uint32_t *__restrict__ parent_of_leaf,
                                                    uint32_t *__restrict__ depth)
{
    const int p = (int)(blockIdx.x * 256u + threadIdx.x);
    if (p >= n || n == 1) return;
    uint32_t levels = 0;
    for (uint32_t node = parent_of_leaf[p]; node != 0xFFFFFFFFu; node = parent_of_node[node])
        if (last[node] - first[node] + 1u > max_leaf) levels++;
    for (int off = 32; off > 0; off >>= 1) levels = max(levels, (uint32_t)__shfl_xor((int)levels, off, 64));
    if ((threadIdx.x & 63u) == 0) atomicMax(depth, levels);
}

} // namespace

hipError_t lbvh_build(const float *h_tri9, const int *h_mat, uint32_t n, const Box &sphere_union, bool have_spheres, const LbvhSphereInput *spheres,
                      uint32_t nspheres, uint32_t max_leaf, uint32_t tree_kind, LbvhOutput &out, hipStream_t stream)
{
    float *d_tri9 = nullptr; int *d_mat = nullptr; Box6 *d_refbox = nullptr, *d_leafbox = nullptr, *d_nodebox = nullptr;
    uint32_t *d_red = nullptr, *d_code = nullptr, *d_idx = nullptr, *d_code_s = nullptr, *d_idx_s = nullptr;
    uint32_t *d_child0 = nullptr, *d_child1 = nullptr, *d_first = nullptr, *d_last = nullptr, *d_pnode = nullptr, *d_pleaf = nullptr, *d_flags = nullptr;
    uint32_t *d_rank = nullptr, *d_depth = nullptr;
    unsigned long long *d_code64 = nullptr, *d_code64_s = nullptr;
    void *d_temp64 = nullptr;
    q4 *d_nodes = nullptr, *d_slots = nullptr, *d_nodes_bf = nullptr, *d_normals = nullptr;
    void *d_temp = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        void *ptrs[] = { d_tri9, d_mat, d_refbox, d_leafbox, d_nodebox, d_red, d_code, d_idx, d_code_s, d_idx_s, d_child0, d_child1, d_first, d_last, d_pnode,
                         d_pleaf, d_flags, d_rank, d_depth, d_nodes, d_slots, d_temp, d_code64, d_code64_s, d_temp64, d_nodes_bf, d_normals };
        for (void *p : ptrs) if (p) (void)hipFree(p);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    if (n == 0 || nspheres > 8) { return hipErrorInvalidValue; }
    const uint32_t blocks = (n + 255u) / 256u, nint = n > 1 ? n - 1 : 1;
    LBVH_TRY(hipMalloc((void **)&d_tri9, (size_t)n * 9 * sizeof(float)));
    LBVH_TRY(hipMalloc((void **)&d_mat, (size_t)n * sizeof(int)));
    LBVH_TRY(hipMalloc((void **)&d_refbox, (size_t)n * sizeof(Box6)));
    LBVH_TRY(hipMalloc((void **)&d_leafbox, (size_t)n * sizeof(Box6)));
    LBVH_TRY(hipMalloc((void **)&d_nodebox, (size_t)nint * sizeof(Box6)));
    LBVH_TRY(hipMalloc((void **)&d_red, 8 * sizeof(uint32_t)));
    uint32_t **u32s[] = { &d_code, &d_idx, &d_code_s, &d_idx_s, &d_pleaf, &d_rank };
    for (uint32_t **pp : u32s) LBVH_TRY(hipMalloc((void **)pp, (size_t)n * sizeof(uint32_t)));
    uint32_t **u32i[] = { &d_child0, &d_child1, &d_first, &d_last, &d_pnode, &d_flags };
    for (uint32_t **pp : u32i) LBVH_TRY(hipMalloc((void **)pp, (size_t)nint * sizeof(uint32_t)));
    LBVH_TRY(hipMalloc((void **)&d_depth, sizeof(uint32_t)));
    LBVH_TRY(hipMalloc((void **)&d_code64, (size_t)n * sizeof(unsigned long long)));
    LBVH_TRY(hipMalloc((void **)&d_code64_s, (size_t)n * sizeof(unsigned long long)));
    LBVH_TRY(hipMalloc((void **)&d_nodes, ((size_t)n + 1) * 4 * sizeof(q4))); // (the SAH builder: n - 1 records + the one above the root)
    LBVH_TRY(hipMalloc((void **)&d_slots, (size_t)n * 3 * sizeof(q4)));
    LBVH_TRY(hipEventCreate(&e0));
    LBVH_TRY(hipEventCreate(&e1));
    LBVH_TRY(hipMemcpyAsync(d_tri9, h_tri9, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice, stream));
    LBVH_TRY(hipMemcpyAsync(d_mat, h_mat, (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
    const uint32_t red_init[8] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u };
    LBVH_TRY(hipMemcpyAsync(d_red, red_init, sizeof red_init, hipMemcpyHostToDevice, stream));
    LBVH_TRY(hipMemsetAsync(d_flags, 0, (size_t)nint * sizeof(uint32_t), stream));
    LBVH_TRY(hipMemsetAsync(d_depth, 0, sizeof(uint32_t), stream));

    LBVH_TRY(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(tri_boxes_kernel, dim3(blocks < 2048u ? blocks : 2048u), dim3(256), 0, stream, d_tri9, n, d_refbox, d_red);
    LBVH_TRY(hipGetLastError());
    uint32_t red[8];
    LBVH_TRY(hipMemcpyAsync(red, d_red, sizeof red, hipMemcpyDeviceToHost, stream));
    LBVH_TRY(hipStreamSynchronize(stream));
    // scene box over ALL primitives (scene.h:167-170): the ground sphere takes part in the normalisation (F10)
    Box scene;
    scene.mn = mk3(ord2f(red[0]), ord2f(red[1]), ord2f(red[2]));
    scene.mx = mk3(ord2f(red[3]), ord2f(red[4]), ord2f(red[5]));
    if (have_spheres) { // interval.h:11-14 per axis
        scene.mn.x = scene.mn.x <= sphere_union.mn.x ? scene.mn.x : sphere_union.mn.x; scene.mx.x = scene.mx.x >= sphere_union.mx.x ? scene.mx.x : sphere_union.mx.x;
        scene.mn.y = scene.mn.y <= sphere_union.mn.y ? scene.mn.y : sphere_union.mn.y; scene.mx.y = scene.mx.y >= sphere_union.mx.y ? scene.mx.y : sphere_union.mx.y;
        scene.mn.z = scene.mn.z <= sphere_union.mn.z ? scene.mn.z : sphere_union.mn.z; scene.mx.z = scene.mx.z >= sphere_union.mx.z ? scene.mx.z : sphere_union.mx.z;
    }
    const f3 extent = scene.mx - scene.mn;
    float maxabs;
    memcpy(&maxabs, &red[6], 4);
    const float margin = fmaxf(1e-5f, maxabs * 3.814697265625e-06f); // as crt_bvh.cpp
    // the spheres' own codes (host, a handful), needed for the triangles' global ranks
    LbvhSphereCodes sc{};
    sc.count = nspheres;
    for (uint32_t k = 0; k < nspheres; k++) sc.code[k] = lbvh_host_morton(spheres[k].refbox, scene);

    out.sorted_codes.resize(n);
    hipLaunchKernelGGL(tri_codes_kernel, dim3(blocks), dim3(256), 0, stream, d_refbox, n, scene.mn.x, scene.mn.y, scene.mn.z, extent.x, extent.y, extent.z,
                       d_code, d_idx);
    LBVH_TRY(hipGetLastError());
    size_t temp_bytes = 0;
    LBVH_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_code, d_code_s, d_idx, d_idx_s, (size_t)n, 0u, 32u, stream));
    LBVH_TRY(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 16));
    LBVH_TRY(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_code, d_code_s, d_idx, d_idx_s, (size_t)n, 0u, 32u, stream));
    hipLaunchKernelGGL(ranks_kernel, dim3(blocks), dim3(256), 0, stream, d_code_s, d_idx_s, n, sc, d_rank);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipMemcpyAsync(out.sorted_codes.data(), d_code_s, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    // second sort, by the tree's own codes (d_code / d_idx are free again)
    {
        const f3 tmn = mk3(ord2f(red[0]), ord2f(red[1]), ord2f(red[2])), tmx = mk3(ord2f(red[3]), ord2f(red[4]), ord2f(red[5]));
        const f3 e = tmx - tmn;
        const float cells = 2097152.0f;
        // cubic cells (one scale for the three axes): a flat scene then simply has no high bits on its thin axis, instead of
        // top-level splits that cut it into sheets (measured on the 1M-triangle grid: 162 ms -> see DESIGN.md)
        const float emax = fmaxf(e.x, fmaxf(e.y, e.z)), scale = emax > 0 ? cells / emax : 0.0f;
        hipLaunchKernelGGL(build_codes_kernel, dim3(blocks), dim3(256), 0, stream, d_refbox, n, tmn.x, tmn.y, tmn.z, scale, scale, scale, d_code64, d_idx);
        LBVH_TRY(hipGetLastError());
        size_t temp64_bytes = 0;
        LBVH_TRY(rocprim::radix_sort_pairs(nullptr, temp64_bytes, d_code64, d_code64_s, d_idx, d_idx_s, (size_t)n, 0u, 63u, stream));
        LBVH_TRY(hipMalloc(&d_temp64, temp64_bytes ? temp64_bytes : 16));
        LBVH_TRY(rocprim::radix_sort_pairs(d_temp64, temp64_bytes, d_code64, d_code64_s, d_idx, d_idx_s, (size_t)n, 0u, 63u, stream));
    }
    GsahResult sah;
    if (tree_kind == LBVH_TREE_SAH) {
        const float tri_box[6] = { ord2f(red[0]), ord2f(red[1]), ord2f(red[2]), ord2f(red[3]), ord2f(red[4]), ord2f(red[5]) };
        float tri_cost = 1.3f; // crt_bvh.cpp: Builder::TRI_COST
        if (const char *e = getenv("CRT_SAH_TRI_COST")) tri_cost = (float)atof(e);
        LBVH_TRY(hipMalloc((void **)&d_nodes_bf, (size_t)n * 4 * sizeof(q4)));
        LBVH_TRY(hipMalloc((void **)&d_normals, (size_t)n * sizeof(q4)));
        LBVH_TRY(gsah_build(d_tri9, d_mat, (const float *)d_refbox, d_idx_s, d_rank, n, margin, tri_box, max_leaf, tri_cost, d_nodes, d_nodes_bf, d_slots, d_normals, sah, stream));
    } else {
        if (n > 1) {
            hipLaunchKernelGGL(hierarchy_kernel, dim3((n - 1 + 255u) / 256u), dim3(256), 0, stream, d_code64_s, (int)n, d_child0, d_child1, d_first, d_last, d_pnode, d_pleaf);
            LBVH_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(fit_kernel, dim3(blocks), dim3(256), 0, stream, d_tri9, d_mat, d_refbox, d_idx_s, (int)n, margin, d_child0, d_child1,
                           d_pnode, d_pleaf, d_leafbox, d_nodebox, d_flags, d_slots, d_rank);
        LBVH_TRY(hipGetLastError());
        if (n > 1) {
            hipLaunchKernelGGL(emit_kernel, dim3((n - 1 + 255u) / 256u), dim3(256), 0, stream, (int)n, max_leaf, d_child0, d_child1, d_first, d_last, d_leafbox,
                               d_nodebox, d_nodes);
            LBVH_TRY(hipGetLastError());
            hipLaunchKernelGGL(depth_kernel, dim3(blocks), dim3(256), 0, stream, (int)n, max_leaf, d_first, d_last, d_pnode, d_pleaf, d_depth);
            LBVH_TRY(hipGetLastError());
        }
    }
    LBVH_TRY(hipEventRecord(e1, stream));

    // results back to the host-side scene description (crt_api.cpp uploads what it renders from)
    out.scene_box = scene;
    out.sphere_codes = sc;
    out.rank_of_tri.resize(n);
    const bool keep = out.keep_on_device && tree_kind == LBVH_TREE_SAH;
    out.slots.clear();
    LBVH_TRY(hipMemcpyAsync(out.rank_of_tri.data(), d_rank, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (!keep) {
        out.slots.resize((size_t)n * 3);
        LBVH_TRY(hipMemcpyAsync(out.slots.data(), d_slots, (size_t)n * 3 * sizeof(q4), hipMemcpyDeviceToHost, stream));
    }
    Box6 rootbox{};
    uint32_t depth = 0;
    if (tree_kind == LBVH_TREE_SAH) {
        out.nodes.resize((size_t)sah.num_records * 4);
        if (sah.num_records) LBVH_TRY(hipMemcpyAsync(out.nodes.data(), d_nodes_bf, (size_t)sah.num_records * 4 * sizeof(q4), hipMemcpyDeviceToHost, stream));
        out.normals.clear();
        if (!keep) {
            out.normals.resize(n);
            LBVH_TRY(hipMemcpyAsync(out.normals.data(), d_normals, (size_t)n * sizeof(q4), hipMemcpyDeviceToHost, stream));
        }
        out.breadth_first = true;
        out.root_ref = sah.root_ref;
        LBVH_TRY(hipStreamSynchronize(stream));
        out.root_box = sah.root_box;
        out.depth = sah.depth;
    } else {
        if (n > max_leaf) {
            out.nodes.resize((size_t)(n - 1) * 4);
            LBVH_TRY(hipMemcpyAsync(out.nodes.data(), d_nodes, (size_t)(n - 1) * 4 * sizeof(q4), hipMemcpyDeviceToHost, stream));
            out.root_ref = 0;
        } else {
            out.nodes.clear();
            out.root_ref = ~(int)((0u << 3) | (n - 1u));
        }
        LBVH_TRY(hipMemcpyAsync(&rootbox, n > 1 ? (const void *)d_nodebox : (const void *)d_leafbox, sizeof rootbox, hipMemcpyDeviceToHost, stream));
        LBVH_TRY(hipMemcpyAsync(&depth, d_depth, sizeof depth, hipMemcpyDeviceToHost, stream));
        LBVH_TRY(hipStreamSynchronize(stream));
        out.root_box.mn = mk3(rootbox.mnx, rootbox.mny, rootbox.mnz);
        out.root_box.mx = mk3(rootbox.mxx, rootbox.mxy, rootbox.mxz);
        out.depth = depth;
    }
    float ms = 0.0f;
    LBVH_TRY(hipEventElapsedTime(&ms, e0, e1));
    out.build_ms = ms;
    out.d_slots = out.d_normals = nullptr;
    if (keep) { out.d_slots = d_slots; out.d_normals = d_normals; d_slots = nullptr; d_normals = nullptr; } // (the caller's from here on)
    cleanup();
    return hipSuccess;
}

} // namespace crt
