// crt_gsah.hip -- binned-SAH construction of the triangle BVH on the GPU (SURVEY section 8 "next" row N1, second builder;
// reference: build_bvh<<<1,1>>>, bvh.h:91-185, one device thread walking a range stack).
//
// The Karras LBVH of crt_lbvh.hip is built in 10 ms and renders 14-26 % slower than the host's binned-SAH tree
// (crt_bvh.cpp: Builder).  This file runs the host builder's own algorithm on the device -- 16 centroid bins per axis, cost
// = area x count on either side, a leaf where splitting does not pay, an object median where the depth bound bites -- so the
// tree has the host tree's quality at a device build's price.  Results cannot differ: boxes only prune (crt_traverse.h).
//
// Two tiers, because a node of a million triangles and a node of five want different machines:
//   * nodes of more than 64 triangles are processed one LEVEL at a time by grid-wide kernels over the triangle positions
//     (a node is a contiguous range of positions, kept so by a stable partition): bins accumulated with LDS tables per
//     work-group that are flushed to the node's bins in device memory (a work-group's 1024 consecutive positions belong to
//     a handful of nodes), one wave per node evaluates the 45 candidate planes, one prefix sum over "goes left" places
//     every triangle, a scatter moves boxes and ids;
//   * a node of at most 64 triangles is finished by ONE wave that holds a triangle per lane and builds the whole sub-tree
//     with ballots, LDS atomics and an LDS permutation, no other wave involved.
// Every node writes its own box and reference into its parent's record (a child knows its exact box from its bins), so no
// bottom-up pass is needed.  Node numbering is allocation order; the host renumbers breadth first as for the other builders.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "crt_bvh.h"
#include "crt_lbvh.h"

namespace crt {

namespace {

#define GSAH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return e_; } } while (0)

constexpr uint32_t SMALL = 64;           // a node of at most this many triangles is one wave's
constexpr uint32_t NBINS = 16;
constexpr uint32_t BIN_WORDS = 13;       // box min xyz, box max xyz, centroid min xyz, centroid max xyz (order-preserving uints), count
constexpr uint32_t NODE_WORDS = 3 * NBINS * BIN_WORDS; // 624
constexpr uint32_t RUNS = 16;            // nodes with an LDS table per work-group of the binning kernel (1024 positions meet at most 16 nodes of 65+)
constexpr uint32_t CHUNK = 1024;         // positions per work-group of the binning kernel
constexpr uint32_t MAX_LEVELS = CRT_STACK_DEPTH;

struct Box6 { float mnx, mny, mnz, mxx, mxy, mxz; };

struct SahNode {      // a node waiting to be split
    uint32_t lo, hi;  // positions [lo, hi)
    uint32_t parent;  // record that receives this node's box and reference ...
    uint32_t meta;    // ... on side (bit 0); levels of nodes still allowed below and including this one (bits 8..15); depth (bits 16..23)
    float cb[6];      // bounds of the centroids (a superset is harmless: bins are clamped)
};
struct SahSplit {     // what the evaluation decided for an active node
    uint32_t axis_bin; // axis << 8 | last bin of the left side; 0xFFFFFFFF = object median in position order
    uint32_t nleft;
    int child[2];     // index in the next level's list, or -1: the child is small (or nothing to do)
};
struct Counters { uint32_t small, nodes, depth, used; uint32_t active[MAX_LEVELS + 2]; };
constexpr uint32_t DEPTH_WORD = 11;      // q[2].w of a record: unused by the kernels (the children's boxes and references take the rest)

__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : u | 0x80000000u; }
__device__ __forceinline__ float ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? o & 0x7FFFFFFFu : ~o); }

__device__ __forceinline__ uint32_t levels_needed(uint32_t n, uint32_t max_leaf)
{ // levels of nodes a median-split sub-tree over n triangles needs (crt_bvh.cpp: Builder::levels_needed)
    uint32_t l = 1;
    unsigned long long cap = 2ull * max_leaf;
    while (cap < n) { cap *= 2; l++; }
    return l;
}
__device__ __forceinline__ uint32_t bin_of(float c, float mn, float scale)
{
    const int k = (int)((c - mn) * scale);
    return (uint32_t)min(max(k, 0), (int)NBINS - 1);
}
__device__ __forceinline__ float bin_scale(float mn, float mx) { const float e = mx - mn; return e > 0.0f ? (float)NBINS / e : 0.0f; }
__device__ __forceinline__ float half_area(const float *b) // b: min xyz, max xyz
{
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx < 0.0f ? 0.0f : dx * dy + dy * dz + dz * dx;
}
// this node's box and reference into its parent's record: side 0 = [min, ref0][max, .], side 1 = [.., ref1 in q[1].w][min][max]
__device__ __forceinline__ void write_child(q4 *nodes, uint32_t parent, uint32_t side, const float *b, int ref)
{
    float *f = (float *)&nodes[4 * (size_t)parent];
    float *mn = f + 8 * side, *mx = f + 8 * side + 4;
    mn[0] = b[0]; mn[1] = b[1]; mn[2] = b[2]; mx[0] = b[3]; mx[1] = b[4]; mx[2] = b[5];
    ((int *)f)[4 * side + 3] = ref;
}

// position p <- triangle idx[p]: its padded box, its id; every position belongs to the root (active node 0) or to nobody
__global__ __launch_bounds__(256) void gsah_init_kernel(const float *__restrict__ refbox6, const uint32_t *__restrict__ idx, uint32_t n, float margin, int root_active,
                                                        Box6 *__restrict__ pbox, uint32_t *__restrict__ pid, int *__restrict__ node_of)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t t = idx[p];
    const float *r = refbox6 + 6 * (size_t)t;
    Box6 b;
    b.mnx = r[0] - margin; b.mny = r[1] - margin; b.mnz = r[2] - margin; b.mxx = r[3] + margin; b.mxy = r[4] + margin; b.mxz = r[5] + margin;
    pbox[p] = b; pid[p] = t; node_of[p] = root_active;
}

__global__ __launch_bounds__(256) void gsah_clear_bins_kernel(uint32_t *__restrict__ gbins, uint32_t words)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) {
        const uint32_t w = i % BIN_WORDS;
        gbins[i] = (w < 3 || (w >= 6 && w < 9)) ? 0xFFFFFFFFu : 0u;
    }
}

template <class Ptr>
__device__ __forceinline__ void bin_add(Ptr bin, const Box6 &b, float cx, float cy, float cz)
{
    atomicMin(&bin[0], f2ord(b.mnx)); atomicMin(&bin[1], f2ord(b.mny)); atomicMin(&bin[2], f2ord(b.mnz));
    atomicMax(&bin[3], f2ord(b.mxx)); atomicMax(&bin[4], f2ord(b.mxy)); atomicMax(&bin[5], f2ord(b.mxz));
    atomicMin(&bin[6], f2ord(cx)); atomicMin(&bin[7], f2ord(cy)); atomicMin(&bin[8], f2ord(cz));
    atomicMax(&bin[9], f2ord(cx)); atomicMax(&bin[10], f2ord(cy)); atomicMax(&bin[11], f2ord(cz));
    atomicAdd(&bin[12], 1u);
}

// bins of the active nodes: per work-group LDS tables for the first RUNS nodes it meets (node ranges are contiguous, so 1024
// positions meet few), flushed with one atomic per non-empty word; further nodes go to device memory directly
__global__ __launch_bounds__(256) void gsah_bin_kernel(const Box6 *__restrict__ pbox, const int *__restrict__ node_of, uint32_t n,
                                                       const SahNode *__restrict__ nodes, uint32_t *__restrict__ gbins)
{
    __shared__ uint32_t lbins[RUNS * NODE_WORDS];
    __shared__ int run_node[RUNS];
    for (uint32_t i = threadIdx.x; i < RUNS * NODE_WORDS; i += 256u) {
        const uint32_t w = i % BIN_WORDS;
        lbins[i] = (w < 3 || (w >= 6 && w < 9)) ? 0xFFFFFFFFu : 0u;
    }
    if (threadIdx.x < RUNS) run_node[threadIdx.x] = -1;
    __syncthreads();
    const uint32_t base = blockIdx.x * CHUNK;
    for (uint32_t it = 0; it < CHUNK / 256u; it++) {
        const uint32_t p = base + it * 256u + threadIdx.x;
        if (p >= n) continue;
        const int a = node_of[p];
        if (a < 0) continue;
        int slot = -1;
        for (uint32_t r = 0; r < RUNS; r++) {
            int cur = run_node[r];
            if (cur == -1) { const int old = atomicCAS(&run_node[r], -1, a); cur = old == -1 ? a : old; }
            if (cur == a) { slot = (int)r; break; }
        }
        const Box6 b = pbox[p];
        const float *cb = nodes[a].cb;
        const float cx = 0.5f * (b.mnx + b.mxx), cy = 0.5f * (b.mny + b.mxy), cz = 0.5f * (b.mnz + b.mxz);
        const uint32_t kx = bin_of(cx, cb[0], bin_scale(cb[0], cb[3])), ky = bin_of(cy, cb[1], bin_scale(cb[1], cb[4])), kz = bin_of(cz, cb[2], bin_scale(cb[2], cb[5]));
        if (slot >= 0) {
            uint32_t *t = lbins + (uint32_t)slot * NODE_WORDS;
            bin_add(t + (0 * NBINS + kx) * BIN_WORDS, b, cx, cy, cz);
            bin_add(t + (1 * NBINS + ky) * BIN_WORDS, b, cx, cy, cz);
            bin_add(t + (2 * NBINS + kz) * BIN_WORDS, b, cx, cy, cz);
        } else {
            uint32_t *t = gbins + (size_t)a * NODE_WORDS;
            bin_add(t + (0 * NBINS + kx) * BIN_WORDS, b, cx, cy, cz);
            bin_add(t + (1 * NBINS + ky) * BIN_WORDS, b, cx, cy, cz);
            bin_add(t + (2 * NBINS + kz) * BIN_WORDS, b, cx, cy, cz);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < RUNS * NODE_WORDS; i += 256u) {
        const uint32_t r = i / NODE_WORDS, w = i % NODE_WORDS, word = w % BIN_WORDS;
        const int a = run_node[r];
        if (a < 0) continue;
        if (lbins[i - word + 12] == 0u) continue; // empty bin
        uint32_t *g = gbins + (size_t)a * NODE_WORDS + w;
        const uint32_t v = lbins[i];
        if (word == 12) atomicAdd(g, v);
        else if (word < 3 || (word >= 6 && word < 9)) atomicMin(g, v);
        else atomicMax(g, v);
    }
}

// wave reduction: lane holding the smallest key
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
        const unsigned long long o = (unsigned long long)hi << 32 | lo;
        v = o < v ? o : v;
    }
    return v;
}

// The 45 candidate planes of a node from its 3 x 16 bins (any uint32 array of NODE_WORDS words).  Lane (axis, k) sweeps its
// axis: left = bins 0..k, right = the others.  Returns the winning lane (or -1: no plane separates anything), and in every lane
// its own left / right boxes, centroid bounds and counts.  Lane 15 (axis 0, k = 15) has everything on its left: the node's box.
struct Sweep { float lbox[6], rbox[6], lcb[6], rcb[6]; uint32_t lcnt, rcnt; float cost; };
template <class Ptr>
__device__ __forceinline__ int evaluate_bins(Ptr bins, const float *cb, int lane, Sweep &s)
{
    const int axis = lane < 48 ? lane / 16 : 0, k = lane & 15;
    uint32_t lo[12], ro[12];
    for (int w = 0; w < 12; w++) { const bool is_min = w < 3 || (w >= 6 && w < 9); lo[w] = ro[w] = is_min ? 0xFFFFFFFFu : 0u; }
    s.lcnt = s.rcnt = 0;
    for (int j = 0; j < (int)NBINS; j++) {
        const uint32_t *b = (const uint32_t *)bins + ((uint32_t)axis * NBINS + (uint32_t)j) * BIN_WORDS;
        const uint32_t cnt = b[12];
        if (cnt == 0) continue;
        if (j <= k) {
            for (int w = 0; w < 12; w++) { const bool is_min = w < 3 || (w >= 6 && w < 9); lo[w] = is_min ? min(lo[w], b[w]) : max(lo[w], b[w]); }
            s.lcnt += cnt;
        } else {
            for (int w = 0; w < 12; w++) { const bool is_min = w < 3 || (w >= 6 && w < 9); ro[w] = is_min ? min(ro[w], b[w]) : max(ro[w], b[w]); }
            s.rcnt += cnt;
        }
    }
    for (int w = 0; w < 6; w++) { s.lbox[w] = ord2f(lo[w]); s.rbox[w] = ord2f(ro[w]); s.lcb[w] = ord2f(lo[6 + w]); s.rcb[w] = ord2f(ro[6 + w]); }
    const bool valid = lane < 48 && k < 15 && s.lcnt > 0 && s.rcnt > 0 && (cb[3 + axis] - cb[axis]) > 0.0f;
    s.cost = valid ? half_area(s.lbox) * (float)s.lcnt + half_area(s.rbox) * (float)s.rcnt : INFINITY;
    const unsigned long long key = (unsigned long long)__float_as_uint(s.cost) << 32 | (uint32_t)lane; // costs are >= 0: bit order = value order
    const unsigned long long best = wave_min_u64(key);
    return (uint32_t)(best >> 32) == 0x7F800000u ? -1 : (int)(uint32_t)best;
}

// one wave per active node: plane, children, this node's record.  Every node of a level splits (a node of more than 64
// triangles is no leaf), so its record is rec_base + its index and its depth is the level's: no counter involved.  The
// children's places in the next level's list / the list of small nodes come from ONE atomic per list and work-group (eight
// nodes): 15,000 nodes each bumping the same five words took 350 us of a level, atomics on one address being served ~5 ns apart.
constexpr uint32_t EVAL_WAVES = 8;
__global__ __launch_bounds__(64 * EVAL_WAVES) void gsah_eval_kernel(const SahNode *__restrict__ nodes, uint32_t nactive, const uint32_t *__restrict__ gbins, uint32_t level,
                                                                    uint32_t rec_base, uint32_t max_leaf, SahSplit *__restrict__ splits, SahNode *__restrict__ next,
                                                                    SahNode *__restrict__ small, q4 *__restrict__ records, Counters *__restrict__ ctr)
{
    __shared__ uint32_t staged[EVAL_WAVES][NODE_WORDS];
    __shared__ uint32_t want[EVAL_WAVES][2], base[2];
    const uint32_t wave = threadIdx.x / 64u, a = blockIdx.x * EVAL_WAVES + wave;
    const int lane = (int)(threadIdx.x & 63u);
    const bool live = a < nactive;
    SahNode nd{};
    if (live) nd = nodes[a];
    const uint32_t n = nd.hi - nd.lo, side = nd.meta & 1u, levels = (nd.meta >> 8) & 0xFFu, depth = (nd.meta >> 16) & 0xFFu;
    // the node's bins through LDS: ten coalesced loads per lane instead of 16 x 13 dependent ones
    uint32_t *mine = staged[wave];
    if (live) for (uint32_t i = (uint32_t)lane; i < NODE_WORDS; i += 64u) mine[i] = gbins[(size_t)a * NODE_WORDS + i];
    __syncthreads();
    float box[6] = { 0, 0, 0, 0, 0, 0 }, ccb[2][6] = { { 0, 0, 0, 0, 0, 0 }, { 0, 0, 0, 0, 0, 0 } };
    uint32_t nleft = n / 2, axis_bin = 0xFFFFFFFFu;
    if (live) {
        Sweep s;
        int win = evaluate_bins(mine, nd.cb, lane, s);
        if (levels <= levels_needed(n, max_leaf)) win = -1; // the depth bound: object median from here on
        for (int w = 0; w < 6; w++) box[w] = __shfl(s.lbox[w], 15, 64);
        if (win >= 0) {
            nleft = (uint32_t)__shfl((int)s.lcnt, win, 64);
            axis_bin = (uint32_t)(win / 16) << 8 | (uint32_t)(win & 15);
            for (int w = 0; w < 6; w++) { ccb[0][w] = __shfl(s.lcb[w], win, 64); ccb[1][w] = __shfl(s.rcb[w], win, 64); }
        } else {
            for (int w = 0; w < 6; w++) ccb[0][w] = ccb[1][w] = nd.cb[w];
        }
    }
    const uint32_t size[2] = { nleft, n - nleft };
    if (lane == 0) {
        want[wave][0] = live ? (size[0] > SMALL ? 1u : 0u) + (size[1] > SMALL ? 1u : 0u) : 0u;
        want[wave][1] = live ? (size[0] > SMALL ? 0u : 1u) + (size[1] > SMALL ? 0u : 1u) : 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t0 = 0, t1 = 0;
        for (uint32_t k = 0; k < EVAL_WAVES; k++) { t0 += want[k][0]; t1 += want[k][1]; }
        base[0] = t0 ? atomicAdd(&ctr->active[level + 1], t0) : 0u;
        base[1] = t1 ? atomicAdd(&ctr->small, t1) : 0u;
    }
    __syncthreads();
    if (!live || lane != 0) return;
    uint32_t at[2] = { base[0], base[1] };
    for (uint32_t k = 0; k < wave; k++) { at[0] += want[k][0]; at[1] += want[k][1]; }
    const uint32_t rec = rec_base + a;
    write_child(records, nd.parent, side, box, (int)rec);
    ((uint32_t *)&records[4 * (size_t)rec])[DEPTH_WORD] = depth; // (a free word of the record: the breadth-first renumbering sorts by it)
    SahSplit sp;
    sp.axis_bin = axis_bin; sp.nleft = nleft;
    for (uint32_t k = 0; k < 2; k++) {
        SahNode c;
        c.lo = k == 0 ? nd.lo : nd.lo + nleft; c.hi = k == 0 ? nd.lo + nleft : nd.hi;
        c.parent = rec; c.meta = k | (levels - 1u) << 8 | (depth + 1u) << 16;
        for (int w = 0; w < 6; w++) c.cb[w] = ccb[k][w];
        if (size[k] > SMALL) { next[at[0]] = c; sp.child[k] = (int)at[0]++; }
        else { small[at[1]++] = c; sp.child[k] = -1; }
    }
    splits[a] = sp;
}

__global__ __launch_bounds__(256) void gsah_flag_kernel(const Box6 *__restrict__ pbox, const int *__restrict__ node_of, uint32_t n, const SahNode *__restrict__ nodes,
                                                        const SahSplit *__restrict__ splits, uint32_t *__restrict__ flag)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const int a = node_of[p];
    uint32_t f = 0;
    if (a >= 0) {
        const uint32_t ab = splits[a].axis_bin;
        if (ab != 0xFFFFFFFFu) {
            const uint32_t axis = ab >> 8;
            const float *cb = nodes[a].cb;
            const float *b = (const float *)&pbox[p];
            const float c = 0.5f * (b[axis] + b[3 + axis]);
            f = bin_of(c, cb[axis], bin_scale(cb[axis], cb[3 + axis])) <= (ab & 0xFFu) ? 1u : 0u;
        }
    }
    flag[p] = f;
}

// stable partition of every split node's range (scan = exclusive prefix sum of flag over all positions)
__global__ __launch_bounds__(256) void gsah_scatter_kernel(const Box6 *__restrict__ pbox, const uint32_t *__restrict__ pid, const int *__restrict__ node_of, uint32_t n,
                                                           const SahNode *__restrict__ nodes, const SahSplit *__restrict__ splits, const uint32_t *__restrict__ flag,
                                                           const uint32_t *__restrict__ scan, Box6 *__restrict__ pbox_out, uint32_t *__restrict__ pid_out,
                                                           int *__restrict__ node_of_out)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const int a = node_of[p];
    uint32_t dst = p;
    int child = -1;
    if (a >= 0) {
        const SahSplit sp = splits[a];
        const uint32_t lo = nodes[a].lo;
        if (sp.axis_bin == 0xFFFFFFFFu) {
            child = sp.child[p - lo < sp.nleft ? 0 : 1];
        } else {
            const uint32_t lefts_before = scan[p] - scan[lo];
            if (flag[p]) { dst = lo + lefts_before; child = sp.child[0]; }
            else { dst = lo + sp.nleft + (p - lo - lefts_before); child = sp.child[1]; }
        }
    }
    pbox_out[dst] = pbox[p]; pid_out[dst] = pid[p]; node_of_out[dst] = child;
}

// One wave finishes a node of at most 64 triangles: a triangle per lane, lanes kept in position order, sub-nodes are lane ranges.
__global__ __launch_bounds__(64) void gsah_small_kernel(const SahNode *__restrict__ small, const Box6 *__restrict__ pbox, uint32_t *__restrict__ pid, uint32_t max_leaf,
                                                        float tri_cost, q4 *__restrict__ records, Counters *__restrict__ ctr)
{
    __shared__ uint32_t bins[NODE_WORDS];
    __shared__ uint32_t cbw[6];
    __shared__ float xch[64 * 7];
    __shared__ uint32_t stack[64 * 3];
    const SahNode nd = small[blockIdx.x];
    const int lane = (int)threadIdx.x;
    const uint32_t n = nd.hi - nd.lo;
    float b[6] = { 0, 0, 0, 0, 0, 0 };
    uint32_t id = 0;
    if ((uint32_t)lane < n) {
        const float *src = (const float *)&pbox[nd.lo + (uint32_t)lane];
        for (int w = 0; w < 6; w++) b[w] = src[w];
        id = pid[nd.lo + (uint32_t)lane];
    }
    uint32_t rec_next = 0;
    if (lane == 0 && n > 1) rec_next = atomicAdd(&ctr->nodes, n - 1u);
    rec_next = (uint32_t)__shfl((int)rec_next, 0, 64);
    uint32_t deepest = 0, used = 0;
    int sp = 0;
    if (lane == 0) { stack[0] = 0u | n << 8; stack[1] = nd.parent; stack[2] = nd.meta; }
    sp = 1;
    __syncthreads();
    while (sp > 0) {
        sp--;
        const uint32_t range = stack[3 * sp], parent = stack[3 * sp + 1], meta = stack[3 * sp + 2];
        const uint32_t lo = range & 0xFFu, hi = range >> 8, m = hi - lo, side = meta & 1u, levels = (meta >> 8) & 0xFFu, depth = (meta >> 16) & 0xFFu;
        const bool in = (uint32_t)lane >= lo && (uint32_t)lane < hi;
        if (m == 1) { // (crt_bvh.cpp: Builder::build, n == 1)
            if (in) write_child(records, parent, side, b, ~(int)(((nd.lo + lo) << 3) | 0u));
            continue;
        }
        __syncthreads(); // (every lane has read the entry popped above before anyone pushes over it)
        for (uint32_t i = (uint32_t)lane; i < NODE_WORDS; i += 64u) {
            const uint32_t w = i % BIN_WORDS;
            bins[i] = (w < 3 || (w >= 6 && w < 9)) ? 0xFFFFFFFFu : 0u;
        }
        if (lane < 6) cbw[lane] = lane < 3 ? 0xFFFFFFFFu : 0u;
        __syncthreads();
        const float cx = 0.5f * (b[0] + b[3]), cy = 0.5f * (b[1] + b[4]), cz = 0.5f * (b[2] + b[5]);
        if (in) {
            atomicMin(&cbw[0], f2ord(cx)); atomicMin(&cbw[1], f2ord(cy)); atomicMin(&cbw[2], f2ord(cz));
            atomicMax(&cbw[3], f2ord(cx)); atomicMax(&cbw[4], f2ord(cy)); atomicMax(&cbw[5], f2ord(cz));
        }
        __syncthreads();
        float cb[6];
        for (int w = 0; w < 6; w++) cb[w] = ord2f(cbw[w]);
        const uint32_t kx = bin_of(cx, cb[0], bin_scale(cb[0], cb[3])), ky = bin_of(cy, cb[1], bin_scale(cb[1], cb[4])), kz = bin_of(cz, cb[2], bin_scale(cb[2], cb[5]));
        if (in) {
            Box6 bb; bb.mnx = b[0]; bb.mny = b[1]; bb.mnz = b[2]; bb.mxx = b[3]; bb.mxy = b[4]; bb.mxz = b[5];
            bin_add(bins + (0 * NBINS + kx) * BIN_WORDS, bb, cx, cy, cz);
            bin_add(bins + (1 * NBINS + ky) * BIN_WORDS, bb, cx, cy, cz);
            bin_add(bins + (2 * NBINS + kz) * BIN_WORDS, bb, cx, cy, cz);
        }
        __syncthreads();
        Sweep s;
        int win = evaluate_bins(bins, cb, lane, s);
        float box[6];
        for (int w = 0; w < 6; w++) box[w] = __shfl(s.lbox[w], 15, 64);
        const bool force_median = levels <= levels_needed(m, max_leaf);
        bool leaf = m <= max_leaf && levels == 0;
        if (!leaf && !force_median) {
            if (win >= 0) { // (crt_bvh.cpp: split_cost against leaf_cost)
                const float best = __shfl(s.cost, win, 64);
                const float split_cost = 1.0f + tri_cost * best / fmaxf(half_area(box), 1e-30f);
                leaf = m <= max_leaf && tri_cost * (float)m <= split_cost;
            } else {
                leaf = m <= max_leaf; // all centroids coincide
            }
        }
        if (leaf) {
            if (lane == 0) write_child(records, parent, side, box, ~(int)(((nd.lo + lo) << 3) | (m - 1u)));
            continue;
        }
        if (force_median) win = -1;
        const uint32_t rec = rec_next++, below_levels = levels > 0 ? levels - 1u : 0u;
        deepest = max(deepest, depth);
        if (lane == 0) { write_child(records, parent, side, box, (int)rec); ((uint32_t *)&records[4 * (size_t)rec])[DEPTH_WORD] = depth; }
        used++;
        uint32_t nleft = m / 2;
        bool left = in && (uint32_t)lane - lo < nleft;
        if (win >= 0) {
            nleft = (uint32_t)__shfl((int)s.lcnt, win, 64);
            const uint32_t axis = (uint32_t)win / 16u, kbest = (uint32_t)win & 15u;
            left = in && (axis == 0 ? kx : axis == 1 ? ky : kz) <= kbest;
        }
        // stable partition of the lanes lo..hi-1 through LDS
        const unsigned long long lmask = __ballot(left), rmask = __ballot(in && !left), below = (1ull << lane) - 1ull;
        if (in) {
            const uint32_t dst = left ? lo + (uint32_t)__popcll(lmask & below) : lo + nleft + (uint32_t)__popcll(rmask & below);
            for (int w = 0; w < 6; w++) xch[7 * dst + (uint32_t)w] = b[w];
            xch[7 * dst + 6] = __uint_as_float(id);
        }
        if (lane == 0) { // right below left on the stack: the left child comes next (depth first keeps the stack short)
            stack[3 * sp] = (lo + nleft) | hi << 8; stack[3 * sp + 1] = rec; stack[3 * sp + 2] = 1u | below_levels << 8 | (depth + 1u) << 16;
            stack[3 * sp + 3] = lo | (lo + nleft) << 8; stack[3 * sp + 4] = rec; stack[3 * sp + 5] = 0u | below_levels << 8 | (depth + 1u) << 16;
        }
        sp += 2;
        __syncthreads();
        if (in) {
            for (int w = 0; w < 6; w++) b[w] = xch[7 * (uint32_t)lane + (uint32_t)w];
            id = __float_as_uint(xch[7 * (uint32_t)lane + 6]);
        }
    }
    if ((uint32_t)lane < n) pid[nd.lo + (uint32_t)lane] = id;
    if (lane == 0 && deepest > 0) { atomicMax(&ctr->depth, deepest); atomicAdd(&ctr->used, used); }
}

// the 48-byte slots in final position order
__global__ __launch_bounds__(256) void gsah_slots_kernel(const float *__restrict__ tri9, const int *__restrict__ mat, const uint32_t *__restrict__ rank_of_tri,
                                                         const uint32_t *__restrict__ pid, uint32_t n, q4 *__restrict__ slots, q4 *__restrict__ normals)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t t = pid[p];
    const float *v = tri9 + 9 * (size_t)t;
    q4 a, b, c;
    a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = __int_as_float((int)t);
    b.x = v[3]; b.y = v[4]; b.z = v[5]; b.w = __int_as_float((int)rank_of_tri[t]);
    c.x = v[6]; c.y = v[7]; c.z = v[8]; c.w = __int_as_float(mat[t]);
    slots[3 * (size_t)p + 0] = a; slots[3 * (size_t)p + 1] = b; slots[3 * (size_t)p + 2] = c;
    normals[p] = triangle_normal(a, b, c); // the hit record's normal (triangle.h:81), the host's operations (crt_traverse.h), once per slot
}

// Breadth-first numbering on the device: records sorted by depth (stable: a level keeps its allocation order), so that "index
// < N" means "one of the N nodes nearest the root" as renumber_breadth_first (crt_bvh.cpp) guarantees for the host's trees.
__global__ __launch_bounds__(256) void gsah_depth_keys_kernel(const q4 *__restrict__ records, uint32_t nrec, uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nrec) return;
    const uint32_t d = ((const uint32_t *)&records[4 * (size_t)i])[DEPTH_WORD];
    key[i] = d ? d : 63u; // records a small node reserved and did not need: behind everything
    val[i] = i;
}
__global__ __launch_bounds__(256) void gsah_inverse_kernel(const uint32_t *__restrict__ order, uint32_t nrec, uint32_t *__restrict__ renum)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j < nrec) renum[order[j]] = j;
}
__global__ __launch_bounds__(256) void gsah_renumber_kernel(const q4 *__restrict__ records, const uint32_t *__restrict__ order, const uint32_t *__restrict__ renum,
                                                            uint32_t used, q4 *__restrict__ out)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= used) return;
    const q4 *src = &records[4 * (size_t)order[j]];
    q4 q[4] = { src[0], src[1], src[2], src[3] };
    int r0 = __float_as_int(q[0].w), r1 = __float_as_int(q[1].w);
    if (r0 >= 0) r0 = (int)renum[r0];
    if (r1 >= 0) r1 = (int)renum[r1];
    q[0].w = __int_as_float(r0); q[1].w = __int_as_float(r1); q[2].w = 0.0f; q[3].w = 0.0f;
    for (int k = 0; k < 4; k++) out[4 * (size_t)j + k] = q[k];
}

} // namespace

hipError_t gsah_build(const float *d_tri9, const int *d_mat, const float *d_refbox6, const uint32_t *d_order, const uint32_t *d_rank_of_tri, uint32_t n,
                      float margin, const float tri_box[6], uint32_t max_leaf, float tri_cost, q4 *d_records, q4 *d_nodes_out, q4 *d_slots, q4 *d_normals, GsahResult &res,
                      hipStream_t stream)
{
    Box6 *pbox[2] = { nullptr, nullptr };
    uint32_t *pid[2] = { nullptr, nullptr }, *d_flag = nullptr, *d_scan = nullptr, *d_gbins = nullptr;
    int *node_of[2] = { nullptr, nullptr };
    SahNode *d_nodes[2] = { nullptr, nullptr }, *d_small = nullptr;
    SahSplit *d_splits = nullptr;
    Counters *d_ctr = nullptr;
    void *d_temp = nullptr;
    auto cleanup = [&]() {
        void *ptrs[] = { pbox[0], pbox[1], pid[0], pid[1], d_flag, d_scan, d_gbins, node_of[0], node_of[1], d_nodes[0], d_nodes[1], d_small, d_splits, d_ctr, d_temp };
        for (void *p : ptrs) if (p) (void)hipFree(p);
    };
    if (n == 0 || max_leaf < 1 || max_leaf > 8) return hipErrorInvalidValue;
    const uint32_t blocks = (n + 255u) / 256u, max_active = n / (SMALL + 1u) + 2u;
    for (int k = 0; k < 2; k++) {
        GSAH_TRY(hipMalloc((void **)&pbox[k], (size_t)n * sizeof(Box6)));
        GSAH_TRY(hipMalloc((void **)&pid[k], (size_t)n * sizeof(uint32_t)));
        GSAH_TRY(hipMalloc((void **)&node_of[k], (size_t)n * sizeof(int)));
        GSAH_TRY(hipMalloc((void **)&d_nodes[k], (size_t)max_active * sizeof(SahNode)));
    }
    GSAH_TRY(hipMalloc((void **)&d_flag, (size_t)n * sizeof(uint32_t)));
    GSAH_TRY(hipMalloc((void **)&d_scan, (size_t)n * sizeof(uint32_t)));
    GSAH_TRY(hipMalloc((void **)&d_gbins, (size_t)max_active * NODE_WORDS * sizeof(uint32_t)));
    GSAH_TRY(hipMalloc((void **)&d_small, (size_t)n * sizeof(SahNode)));
    GSAH_TRY(hipMalloc((void **)&d_splits, (size_t)max_active * sizeof(SahSplit)));
    GSAH_TRY(hipMalloc((void **)&d_ctr, sizeof(Counters)));
    size_t temp_bytes = 0;
    GSAH_TRY(rocprim::exclusive_scan(nullptr, temp_bytes, d_flag, d_scan, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    GSAH_TRY(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 16));

    GSAH_TRY(hipMemsetAsync(d_ctr, 0, sizeof(Counters), stream));
    GSAH_TRY(hipMemsetAsync(d_records, 0, ((size_t)n + 1) * 4 * sizeof(q4), stream));
    // the root: parent = the spare record n ("above the root"), side 0; its centroid bounds are the triangles' box (a superset)
    SahNode root;
    root.lo = 0; root.hi = n; root.parent = n; root.meta = 0u | MAX_LEVELS << 8 | 1u << 16;
    for (int w = 0; w < 6; w++) root.cb[w] = tri_box[w];
    const bool tiered = n > SMALL;
    GSAH_TRY(hipMemcpyAsync(tiered ? d_nodes[0] : d_small, &root, sizeof root, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(gsah_init_kernel, dim3(blocks), dim3(256), 0, stream, d_refbox6, d_order, n, margin, tiered ? 0 : -1, pbox[0], pid[0], node_of[0]);
    GSAH_TRY(hipGetLastError());
    int cur = 0;
    uint32_t nactive = tiered ? 1u : 0u, nsmall = tiered ? 0u : 1u, levels_run = 0, level_records = 0;
    for (uint32_t level = 0; nactive > 0; level++) {
        if (level > MAX_LEVELS) { cleanup(); return hipErrorUnknown; } // (cannot happen: the depth bound forces medians)
        const uint32_t words = nactive * NODE_WORDS;
        hipLaunchKernelGGL(gsah_clear_bins_kernel, dim3(std::min((words + 255u) / 256u, 2048u)), dim3(256), 0, stream, d_gbins, words);
        hipLaunchKernelGGL(gsah_bin_kernel, dim3((n + CHUNK - 1u) / CHUNK), dim3(256), 0, stream, pbox[cur], node_of[cur], n, d_nodes[cur], d_gbins);
        hipLaunchKernelGGL(gsah_eval_kernel, dim3((nactive + EVAL_WAVES - 1u) / EVAL_WAVES), dim3(64 * EVAL_WAVES), 0, stream, d_nodes[cur], nactive, d_gbins, level,
                           level_records, max_leaf, d_splits, d_nodes[cur ^ 1], d_small, d_records, d_ctr);
        level_records += nactive; // (every active node took one record)
        hipLaunchKernelGGL(gsah_flag_kernel, dim3(blocks), dim3(256), 0, stream, pbox[cur], node_of[cur], n, d_nodes[cur], d_splits, d_flag);
        GSAH_TRY(hipGetLastError());
        GSAH_TRY(rocprim::exclusive_scan(d_temp, temp_bytes, d_flag, d_scan, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(gsah_scatter_kernel, dim3(blocks), dim3(256), 0, stream, pbox[cur], pid[cur], node_of[cur], n, d_nodes[cur], d_splits, d_flag, d_scan,
                           pbox[cur ^ 1], pid[cur ^ 1], node_of[cur ^ 1]);
        GSAH_TRY(hipGetLastError());
        Counters h;
        GSAH_TRY(hipMemcpyAsync(&h, d_ctr, sizeof h, hipMemcpyDeviceToHost, stream));
        GSAH_TRY(hipStreamSynchronize(stream));
        nactive = h.active[level + 1]; nsmall = h.small;
        if (nactive > max_active) { cleanup(); return hipErrorUnknown; }
        cur ^= 1;
        levels_run = level + 1;
    }
    // the small nodes take their records behind the levels' (which no counter counted), and count their own depth and use
    GSAH_TRY(hipMemcpyAsync(&d_ctr->nodes, &level_records, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    if (nsmall > 0) {
        hipLaunchKernelGGL(gsah_small_kernel, dim3(nsmall), dim3(64), 0, stream, d_small, pbox[cur], pid[cur], max_leaf, tri_cost, d_records, d_ctr);
        GSAH_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(gsah_slots_kernel, dim3(blocks), dim3(256), 0, stream, d_tri9, d_mat, d_rank_of_tri, pid[cur], n, d_slots, d_normals);
    GSAH_TRY(hipGetLastError());
    Counters h;
    q4 above[4];
    GSAH_TRY(hipMemcpyAsync(&h, d_ctr, sizeof h, hipMemcpyDeviceToHost, stream));
    GSAH_TRY(hipMemcpyAsync(above, d_records + 4 * (size_t)n, sizeof above, hipMemcpyDeviceToHost, stream));
    GSAH_TRY(hipStreamSynchronize(stream));
    h.used += level_records;                               // (the levels' records are all in use)
    h.depth = h.depth > levels_run ? h.depth : levels_run; // (level l holds the nodes of depth l + 1)
    if (h.nodes > 0) { // breadth-first numbering; the tier-A buffers are free by now: keys / values of the sort live in them
        if (h.nodes >= n || h.used > h.nodes) { cleanup(); return hipErrorUnknown; }
        uint32_t *key_in = d_flag, *key_out = d_scan, *val_in = pid[cur ^ 1], *val_out = (uint32_t *)node_of[0], *renum = (uint32_t *)node_of[1];
        const uint32_t rb = (h.nodes + 255u) / 256u;
        hipLaunchKernelGGL(gsah_depth_keys_kernel, dim3(rb), dim3(256), 0, stream, d_records, h.nodes, key_in, val_in);
        size_t sort_bytes = 0;
        void *d_sort = nullptr;
        GSAH_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, key_in, key_out, val_in, val_out, (size_t)h.nodes, 0u, 6u, stream));
        GSAH_TRY(hipMalloc(&d_sort, sort_bytes ? sort_bytes : 16));
        hipError_t e = rocprim::radix_sort_pairs(d_sort, sort_bytes, key_in, key_out, val_in, val_out, (size_t)h.nodes, 0u, 6u, stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gsah_inverse_kernel, dim3(rb), dim3(256), 0, stream, val_out, h.nodes, renum);
            hipLaunchKernelGGL(gsah_renumber_kernel, dim3((h.used + 255u) / 256u), dim3(256), 0, stream, d_records, val_out, renum, h.used, d_nodes_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_sort);
        GSAH_TRY(e);
    }
    res.num_records = h.used; res.depth = h.depth; res.levels_run = levels_run; res.small_nodes = h.small;
    std::memcpy(&res.root_ref, &above[0].w, 4);
    if (res.root_ref >= 0) res.root_ref = 0; // the root is the one record of depth 1: first after the sort
    res.root_box.mn = mk3(above[0].x, above[0].y, above[0].z);
    res.root_box.mx = mk3(above[1].x, above[1].y, above[1].z);
    cleanup();
    return hipSuccess;
}

} // namespace crt
