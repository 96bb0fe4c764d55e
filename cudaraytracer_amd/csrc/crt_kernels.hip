// crt_kernels.hip -- gfx950 kernels of the hot path and their launchers.
//
// render_kernel: persistent wave64 waves.  A lane owns one work unit at a time -- a pixel, or one chunk of a pixel's
// samples -- taken in blocks of 2^fetch_shift units (RenderArgs, chosen by launch_render) from 64 striped block counters (units are numbered in 8x8-tile
// order: the reference's block shape, kernel.cu:50-51, renderer.h:51-52); a unit's samples are summed in the lane in
// sample order (renderer.h:28-32), so the float sum is the reference's whatever the partition.  20 waves per CU in one of
// three forms, chosen per frame by launch_render: one wave per workgroup (launches that do not fill the GPU); five 4-wave
// workgroups that each share an LDS copy of the top of the tree (418 / 236 / 88 nodes beside stacks of 14 / 16 / 24
// levels); or, for frames that fill the GPU many times over, TWO launches on one work queue -- one 16-wave workgroup per CU
// whose LDS copy holds the whole node array of a small tree, and one 4-wave workgroup with what LDS is left (launch_split).
// The traversal stack is an LDS column per lane (stack[level][lane]: bank = lane, conflict free), with 16-bit entries where
// the tree's references fit a short.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <mutex>

#include "crt_kernels.h"
#include "crt_traverse.h"

namespace crt {

// Per-lane traversal stack: an LDS column (stack[level][lane]: bank = lane, conflict free).  Level 0 holds the
// CRT_BVH_NONE sentinel for the whole kernel, so pop() needs no empty check: popping it ends the walk.
struct LdsStack {
    typedef int entry;
    int *top; // next free slot of this lane's column (stride 64 ints)
    int *base;
    __device__ __forceinline__ void init(int *column) { column[0] = CRT_BVH_NONE; base = column + 64; top = base; }
    __device__ __forceinline__ void reset() { top = base; }
    __device__ __forceinline__ void push(int v) { *top = v; top += 64; }
    __device__ __forceinline__ int pop() { top -= 64; return *top; }
    // branch-free forms for the node step: the store always happens (one spare level above the deepest push takes it)
    __device__ __forceinline__ int peek() { return *(top - 64); }
    __device__ __forceinline__ void push_if(int v, bool c) { *top = v; top += c ? 64 : 0; }
    __device__ __forceinline__ void pop_if(bool c) { top -= c ? 64 : 0; }
};

// The same column with 16-bit entries, for trees whose every reference fits a short (fewer than 32767 nodes, at most 4096
// triangle slots: leaf references are ~((first << 3) | (count - 1)) >= -32768): half the LDS per wave, which the copy of the top
// of the tree gets.  The end-of-walk value is 0x7FFF here (it must survive the trip through a short), and so it is for the
// whole kernel that uses this stack: stack_none<> below, used wherever the 32-bit kernels say CRT_BVH_NONE.
struct LdsStack16 {
    typedef short entry;
    short *top, *base;
    __device__ __forceinline__ void init(short *column) { column[0] = (short)0x7FFF; base = column + 64; top = base; }
    __device__ __forceinline__ void reset() { top = base; }
    __device__ __forceinline__ void push(int v) { *top = (short)v; top += 64; }
    __device__ __forceinline__ int pop() { top -= 64; return *top; }
    __device__ __forceinline__ int peek() { return *(top - 64); }
    __device__ __forceinline__ void push_if(int v, bool c) { *top = (short)v; top += c ? 64 : 0; }
    __device__ __forceinline__ void pop_if(bool c) { top -= c ? 64 : 0; }
};
template <> struct stack_none<LdsStack16> { static constexpr int value = 0x7FFF; };

// The same stack with its top entry in a lane register (BASELINE.json's "short stack in VGPRs spilling to LDS", one entry
// deep): peek() costs nothing, but the branch-free node step must still read what a pop would uncover (now the second
// entry) and store what a push covers (now the old top), so the LDS traffic per step is the same one read and one write,
// and two more selects keep the register current.  Measured with -DCRT_REG_TOS=1: slower; kept for
// the record, not used.
struct RegTopStack {
    typedef int entry;
    int *top;  // next free LDS slot of this lane's column: entries [base, top) lie below `tos`
    int *base;
    int tos;   // the top entry
    int spec;  // what a pop would uncover, read ahead by peek()
    __device__ __forceinline__ void init(int *column) { column[0] = CRT_BVH_NONE; base = column + 64; top = base; tos = CRT_BVH_NONE; spec = CRT_BVH_NONE; }
    __device__ __forceinline__ void reset() { top = base; tos = CRT_BVH_NONE; }
    __device__ __forceinline__ void push(int v) { *top = tos; top += 64; tos = v; }
    __device__ __forceinline__ int pop() { const int r = tos; top -= 64; tos = *top; return r; }
    __device__ __forceinline__ int peek() { spec = *(top - 64); return tos; }
    __device__ __forceinline__ void push_if(int v, bool c) { *top = tos; top += c ? 64 : 0; tos = c ? v : tos; }
    __device__ __forceinline__ void pop_if(bool c) { top -= c ? 64 : 0; tos = c ? spec : tos; }
};
#ifndef CRT_REG_TOS
#define CRT_REG_TOS 0
#endif

template <bool SS>
struct RegCounters {
    static constexpr bool sphere_slots = SS;
    uint32_t nodes = 0, tris = 0, spheres = 0, rays = 0, refolds = 0, shades = 0;
    unsigned long long *refold_counter = nullptr; // (unused here: refolds are summed with the other counters)
    __device__ __forceinline__ void node() { nodes++; }
    __device__ __forceinline__ void tri() { tris++; }
    __device__ __forceinline__ void sphere() { spheres++; }
    __device__ __forceinline__ void ray() { rays++; }
    __device__ __forceinline__ void refold() { refolds++; }
    __device__ __forceinline__ void shade() { shades++; }
};
template <bool SS>
struct RayCounters { // always-on: rays and tie refolds only
    static constexpr bool sphere_slots = SS;
    uint32_t rays = 0;
    unsigned long long *refold_counter = nullptr; // tie refolds are a few per million rays: counted where they happen, not in a lane register
    __device__ __forceinline__ void node() {}
    __device__ __forceinline__ void tri() {}
    __device__ __forceinline__ void sphere() {}
    __device__ __forceinline__ void ray() { rays++; }
    __device__ __forceinline__ void refold() { atomicAdd(refold_counter, 1ull); }
    __device__ __forceinline__ void shade() {}
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// renderer.h:36-37 + color.h:5-24: average, gamma (or the raw sum in accumulate mode)
__device__ __forceinline__ float finish_channel(float v, float spp, bool raw)
{
    if (raw) return v;
    v = v / spp;
    return v > 0 ? __builtin_sqrtf(v) : 0.0f;
}
// ... and store, for the kernels whose lanes sum their unit's samples themselves (RenderArgs.chunk_mode CM_WHOLE / CM_SLAB).
// q = where the unit goes: the pixel (or compact slot), or with slabs (pixel or slot) * chunks + chunk -- then `out` is the
// chunk scratch [pixel][chunk][3] and receives the raw chunk sum (finalize_chunks_kernel adds a pixel's chunk sums in chunk order).
__device__ __forceinline__ void store_unit(const RenderArgs &ra, float *__restrict__ out, uint32_t q, f3 col)
{
    const bool raw = (ra.flags & RA_ACCUM) != 0 || ra.chunk_mode == CM_SLAB;
    const size_t idx = q;
    f3 o = col;
    if (!raw) o = mk3(finish_channel(col.x, (float)ra.spp, false), finish_channel(col.y, (float)ra.spp, false), finish_channel(col.z, (float)ra.spp, false));
    out[3 * idx + 0] = o.x; out[3 * idx + 1] = o.y; out[3 * idx + 2] = o.z; // (one twelve-byte store: every vector memory instruction costs the CU's address unit the same)
}

// pixel slot (8x8-tile order over this rank's tiles) -> column | row << 16; false: outside the image (ragged last tile column / row)
__device__ __forceinline__ bool slot_pixel(const RenderArgs &ra, uint32_t slot, uint32_t &ij)
{
    const uint32_t tile = (slot >> 6) * ra.tile_ranks + ra.tile_rank, l = slot & 63u;
    // tile / tiles_x by the host's floor(2^32 / tiles_x): at most one too small
    uint32_t ty = __umulhi(tile, ra.tiles_x_magic), tx = tile - ty * ra.tiles_x;
    if (tx >= ra.tiles_x) { tx -= ra.tiles_x; ty++; }
    const uint32_t i = tx * 8u + (l & 7u), j = ty * 8u + (l >> 3);
    ij = i | (j << 16);
    return i < ra.width && j < ra.height; // renderer.h:22
}

// Sample chunks (RenderArgs.chunks = n > 1): the work unit is ONE SAMPLE of a pixel, and the units of a block of 2^group_shift
// pixels (x spp samples) are rendered by the lanes of ONE wave, which also adds them up: every finished sample goes to the wave's
// own ring in device memory (ring_slots blocks of block_units x 12 bytes, small enough to stay in the L2;
// lanes of one wave see each other's stores without any cache maintenance: a wave's vector memory operations reach its L1 in
// order), a count per ring slot says when a block is complete, and the wave then adds each pixel's n chunk sums IN CHUNK ORDER,
// STARTING FROM 0 -- with one sample per chunk the float operations of the reference's own loop (renderer.h:26-32), so the frame
// is the sequential sum bit for bit -- and writes the finished pixels.  One lane per (pixel of the block, colour channel): the
// three chains of a pixel are independent, a chain is n dependent additions.  No slabs in HBM, no second kernel.
typedef __attribute__((address_space(1))) float *global_f32_ptr;
#ifndef CRT_RING_LOADS_IN_FLIGHT
#define CRT_RING_LOADS_IN_FLIGHT 32
#endif
// `done`: the ring slots whose blocks are complete (at least one).  Their pixels' chains are dealt to the lanes together -- the
// pass costs the same instructions whether 6 or 60 lanes take part, so the caller lets complete blocks collect (RenderArgs.ring_batch).
__device__ __forceinline__ void finalize_ring_blocks(const RenderArgs &ra, float *__restrict__ out, global_f32_ptr wring, uint32_t done,
                                                     const uint32_t *ring_block, uint32_t lane)
{
    unsigned long long list = 0; // the slots of the complete blocks, four bits each
    uint32_t nblocks = 0;
    for (uint32_t m = done; m; m &= m - 1u, nblocks++) list |= (unsigned long long)__builtin_ctz(m) << (4u * nblocks);
    const uint32_t chains = (3u << ra.group_shift) * nblocks, n = ra.chunks;
    const bool raw = (ra.flags & RA_ACCUM) != 0;
    for (uint32_t idx = lane; idx < ((chains + 63u) & ~63u); idx += 64u) { // (wave-uniform trip count)
        if (idx < chains) {
            const uint32_t pix = __umulhi(idx, 0x55555556u), c = idx - 3u * pix; // idx / 3 for idx < 2^31: pixel of the pass, colour channel
            const uint32_t rs = (uint32_t)(list >> (4u * (pix >> ra.group_shift))) & 15u, g = pix & ((1u << ra.group_shift) - 1u);
            const uint32_t slot = (ring_block[rs] << ra.group_shift) + g;
            uint32_t ij;
            const bool inside = slot_pixel(ra, slot, ij);
            global_f32_ptr src = wring + (rs * ra.block_units + g * ra.spp) * 3u + c; // the pixel's samples, 3 floats apart
            float acc = 0.0f; // renderer.h:26
#ifdef CRT_RING_NOSUM // timing probe: what the additions and their loads cost (the frame is wrong)
            acc = src[0];
#else
            if (ra.chunk_spp == 1u) { // one sample per chunk: the reference's own loop (a chunk sum 0 + s is s: the sum so far is never -0)
                uint32_t k = 0;
#if CRT_RING_LOADS_IN_FLIGHT > 8
                // long chains (hundreds of samples per pixel) wait for their loads, not for their additions: more of them in flight
                // (4K x 1024 spp: 521 ms with 32, 528 with 16, 538 with 8; at 64 spp the short form is as fast and holds fewer registers)
                for (; n >= 128u && k + (uint32_t)CRT_RING_LOADS_IN_FLIGHT <= n; k += (uint32_t)CRT_RING_LOADS_IN_FLIGHT, src += 3 * CRT_RING_LOADS_IN_FLIGHT) {
                    float v[CRT_RING_LOADS_IN_FLIGHT];
#pragma unroll
                    for (int j = 0; j < CRT_RING_LOADS_IN_FLIGHT; j++) v[j] = src[3 * j];
#pragma unroll
                    for (int j = 0; j < CRT_RING_LOADS_IN_FLIGHT; j++) acc = acc + v[j];
                }
#endif
                for (; k + 8u <= n; k += 8u, src += 24) { // eight loads in flight, their offsets in the instructions
                    const float s0 = src[0], s1 = src[3], s2 = src[6], s3 = src[9], s4 = src[12], s5 = src[15], s6 = src[18], s7 = src[21];
                    acc = acc + s0; acc = acc + s1; acc = acc + s2; acc = acc + s3; acc = acc + s4; acc = acc + s5; acc = acc + s6; acc = acc + s7;
                }
                for (; k < n; k++, src += 3) acc = acc + src[0];
            } else {
                for (uint32_t j = 0; j < n; j++, src += 3u * ra.chunk_spp) {
                    float chunk_sum = 0.0f;
#pragma unroll 4
                    for (uint32_t k = 0; k < ra.chunk_spp; k++) chunk_sum = chunk_sum + src[3u * k];
                    acc = acc + chunk_sum;
                }
            }
#endif
            const size_t px = (ra.flags & RA_COMPACT) ? slot : (size_t)(ij >> 16) * ra.width + (ij & 0xFFFFu);
            if (inside) out[3 * px + c] = finish_channel(acc, (float)ra.spp, raw);
            else if (ra.flags & RA_COMPACT) out[3 * px + c] = 0.0f;
        }
    }
}

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }
// number of set bits of `mask` below this lane
__device__ __forceinline__ int lane_rank(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// The camera is 34 dwords of kernel argument that only the camera-ray code reads.  Left to the compiler they are loaded
// into scalar registers at kernel entry, do not fit beside everything else and are parked in VGPR lanes: ~80
// v_readlane per shade phase on a kernel that is bound by VALU issue.  Here they are read from the kernel-argument
// segment where they are used (scalar loads through the constant cache); the empty asm keeps the compiler from
// recognising the pointer and hoisting the loads back to the entry.
struct KernelArgsLayout { SceneView sc; CameraPOD cam; RenderArgs ra; float *out; int *ids; unsigned long long *counters; };
template <class T>
__device__ __forceinline__ T load_kernel_arg(size_t offset)
{
    typedef __attribute__((address_space(4))) const unsigned char *kernarg_bytes;
    typedef __attribute__((address_space(4))) const uint32_t *kernarg_words;
    kernarg_bytes base = (kernarg_bytes)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    kernarg_words w = (kernarg_words)(base + offset);
    union { T v; uint32_t u[sizeof(T) / 4]; } r;
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; k++) r.u[k] = w[k];
    return r.v;
}

// Node records: the first `ncached` nodes (breadth-first numbering = the top of the tree) from the workgroup's LDS copy,
// the others from device memory.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4f *lds_q4_ptr; // keeps the LDS copy on ds_read (a generic pointer would load via flat_)
__device__ __forceinline__ q4 to_q4(v4f v) { q4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
struct CachedNodes {
    const q4 *global;
    lds_q4_ptr lds;
    int ncached;
    __device__ __forceinline__ void fetch(int cur, q4 &n0, q4 &n1, q4 &n2, q4 &n3) const
    {
        if (cur < ncached) {
            lds_q4_ptr n = lds + 4u * (uint32_t)cur;
            n0 = to_q4(n[0]); n1 = to_q4(n[1]); n2 = to_q4(n[2]); n3 = to_q4(n[3]);
        } else {
            const q4 *n = global + 4u * (uint32_t)cur;
            n0 = n[0]; n1 = n[1]; n2 = n[2]; n3 = n[3];
        }
    }
};


// Four-child nodes (crt_traverse.h): 112 bytes of a 128-byte device record, the first `ncached` of them from the LDS copy.
#define CRT_LDS_WIDE_WORDS 28u
struct CachedNodes4 {
    const q4 *global;
    lds_q4_ptr lds;
    uint32_t ncached;
    __device__ __forceinline__ void fetch(uint32_t idx, q4 &mnx, q4 &mny, q4 &mnz, q4 &mxx, q4 &mxy, q4 &mxz, q4 &ref) const
    {
        if (idx < ncached) {
            lds_q4_ptr n = lds + 7u * idx;
            mnx = to_q4(n[0]); mny = to_q4(n[1]); mnz = to_q4(n[2]); mxx = to_q4(n[3]); mxy = to_q4(n[4]); mxz = to_q4(n[5]); ref = to_q4(n[6]);
        } else {
            const q4 *n = global + 8u * idx;
            mnx = n[0]; mny = n[1]; mnz = n[2]; mxx = n[3]; mxy = n[4]; mxz = n[5]; ref = n[6];
        }
    }
};

struct LdsNodes { // the LDS copy only (the caller has checked cur < ncached)
    lds_q4_ptr lds;
    __device__ __forceinline__ void fetch(int cur, q4 &n0, q4 &n1, q4 &n2, q4 &n3) const
    {
        lds_q4_ptr n = lds + 4u * (uint32_t)cur;
        n0 = to_q4(n[0]); n1 = to_q4(n[1]); n2 = to_q4(n[2]); n3 = to_q4(n[3]);
#ifdef CRT_EXTRA_MINMAX
        { // cost probe: CRT_EXTRA_MINMAX independent v_min_f32 per LDS node step (results unused)
            float a = n0.x, b = n1.x;
#pragma unroll
            for (int k = 0; k < CRT_EXTRA_MINMAX; k++) asm volatile("v_min_f32_e32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
        }
#endif
    }
};

// The LDS node step with the slab planes picked by the signs of the ray direction instead of by min / max.  A ray whose
// direction is positive along x enters every box through its min.x plane and leaves through max.x (t = plane * idir - ood is
// monotone in the plane, roundings included), so near and far planes are known per ray and need not be found per node: per
// child max3 + max for the entry time and min3 + min for the exit time, 8 instructions of the 4-cycle class per step where the
// pairwise form takes 20.  The price is per-lane addressing -- twelve 4-byte LDS reads at `node + 16 * (sign bit)` instead of
// four 16-byte ones, and six address additions -- which only the LDS copy can afford (from device memory every load costs
// the address unit 16 cycles whatever its width).  Same boxes, same decisions as trav_inner_from, bit for bit.

typedef __attribute__((address_space(3))) const float *lds_f32_ptr;
typedef __attribute__((address_space(3))) const unsigned char *lds_u8_ptr;
// The LDS copy holds a node in 56 bytes, not 64 (the two unused words of the device record are dropped: 14 % more nodes in the
// same LDS): [c0.min xyz][c0.max xyz][c1.min xyz][c1.max xyz][ref0 ref1].
#define CRT_LDS_NODE_BYTES 56u
#define CRT_LDS_NODE_WORDS 14u
struct SignOffsets { uint32_t nx, ny, nz, fx, fy, fz; }; // byte offsets of the near / far plane rows within a node: 0 (min row) or 12 (max row)
__device__ __forceinline__ SignOffsets sign_offsets(const Trav &tv)
{
    SignOffsets o;
    o.nx = ((uint32_t)(f2i(tv.idir.x) >> 31)) & 12u; o.ny = ((uint32_t)(f2i(tv.idir.y) >> 31)) & 12u; o.nz = ((uint32_t)(f2i(tv.idir.z) >> 31)) & 12u;
    o.fx = o.nx ^ 12u; o.fy = o.ny ^ 12u; o.fz = o.nz ^ 12u;
    return o;
}
template <class Stack, class Counters>
__device__ __forceinline__ void trav_inner_lds_sorted(lds_q4_ptr nodes, const SignOffsets &so, Trav &tv, Stack &stack, Counters &cnt)
{
    const lds_u8_ptr node = (lds_u8_ptr)nodes + CRT_LDS_NODE_BYTES * (uint32_t)tv.cur; // (a stride that spreads the nodes over all LDS banks, 68 bytes: no change)
    const lds_f32_ptr nx = (lds_f32_ptr)(node + so.nx), ny = (lds_f32_ptr)(node + so.ny), nz = (lds_f32_ptr)(node + so.nz);
    const lds_f32_ptr fx = (lds_f32_ptr)(node + so.fx), fy = (lds_f32_ptr)(node + so.fy), fz = (lds_f32_ptr)(node + so.fz);
    const lds_f32_ptr row = (lds_f32_ptr)node;
    // child 0: rows at +0 (min) and +12 (max); child 1: +24 and +36, i.e. 6 floats further; the references behind them
    const float n0x = nx[0], n0y = ny[1], n0z = nz[2], f0x = fx[0], f0y = fy[1], f0z = fz[2];
    const float n1x = nx[6], n1y = ny[7], n1z = nz[8], f1x = fx[6], f1y = fy[7], f1z = fz[8];
    const int r0 = f2i(row[12]), r1 = f2i(row[13]);
    const int below = stack.peek();
    cnt.node();
    const f3 idir = tv.idir, ood = tv.ood;
    const float lo = -tv.shift, hi = tv.hi;
    const float tn0 = __builtin_fmaxf(__builtin_fmaxf(fma1(n0x, idir.x, -ood.x), fma1(n0y, idir.y, -ood.y)), __builtin_fmaxf(fma1(n0z, idir.z, -ood.z), lo));
    const float tf0 = __builtin_fminf(__builtin_fminf(fma1(f0x, idir.x, -ood.x), fma1(f0y, idir.y, -ood.y)), min_no_nan(fma1(f0z, idir.z, -ood.z), hi));
    const float tn1 = __builtin_fmaxf(__builtin_fmaxf(fma1(n1x, idir.x, -ood.x), fma1(n1y, idir.y, -ood.y)), __builtin_fmaxf(fma1(n1z, idir.z, -ood.z), lo));
    const float tf1 = __builtin_fminf(__builtin_fminf(fma1(f1x, idir.x, -ood.x), fma1(f1y, idir.y, -ood.y)), min_no_nan(fma1(f1z, idir.z, -ood.z), hi));
    trav_descend(tn0, tf0, tn1, tf1, r0, r1, below, tv, stack);
}

// Section clock of the instrumented kernel (CRT_COUNT_WORK): shader-clock ticks a wave spends in each part of the loop
template <bool ON>
struct SectionClock {
    unsigned long long sum[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, last = 0;
    __device__ __forceinline__ void start() { if constexpr (ON) { __builtin_amdgcn_sched_barrier(0); last = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } }
    template <int K>
    __device__ __forceinline__ void lap()
    {
        if constexpr (ON) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("; crt_section_end %0" ::"n"(K)); // a marker for tools/isa_sections.py
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            sum[K] += now - last; last = now;
        }
    }
};
#define CRT_LAP(clk, k) (clk).template lap<k>()

// Lane phases while not walking (a lane walks the BVH iff tv.cur != CRT_BVH_NONE)
enum { P_STARVED = -1, P_DONE = 0, P_FETCH = 1, P_NEW = 2, P_HIT = 3 }; // (P_STARVED: needs a work unit while the wave's sample-chunk ring is full)

// render_kernel: persistent waves, each on its own (W = 1: one per workgroup; W = 4: four per workgroup that share an LDS
// copy of the top of the tree and otherwise never meet).  A lane owns one work unit at a time: a pixel, its samples summed
// in sample order (renderer.h:28-32, so the float sum is the reference's whatever the partition), or with sample chunks one
// chunk of a pixel's samples; units are numbered in 8x8-tile order and handed out in blocks of 64 from striped counters
// (phase A below), so a lane that finishes does not wait for the slowest lane of its tile.
//
// Divergence control.  Paths end after 1..max_depth segments; 58 % of the segments of the headline scene never enter
// the triangle BVH (they miss its root box), the others walk 14 steps on average with a long tail.  Lock-step "sample
// loop / bounce loop / traversal loop" nesting leaves ~80 % of the lanes idle (profiles/r1_v1_nested_loops).  Here
// every lane is a small state machine
//     FETCH (needs a unit) -> NEW (needs a camera ray) -> walking -> HIT (walk over: resolve, shade, scatter) -> ...
// and the wave alternates, by 64-bit ballots, between two phases:
//   A  "shade", run when at least `sched_threshold` lanes are not walking (or nobody walks): HIT lanes resolve their
//      hit, fetch primitive and shading record together, add emission and end the path or scatter; finished units are
//      stored and new ones fetched; NEW lanes build the next sample's camera ray (one Philox evaluation serves both kinds
//      of lane; defocus-disk rejection loop inline, utility.h:100-109); then the sphere tests and the root-box test
//      start the next walk.
//   B  "walk": by vote, node steps for the lanes at a node (LDS-only steps or device-memory steps, see the loop) or one
//      leaf step for the lanes at a leaf, until enough lanes wait for phase A again.
// A lane's own operation sequence is exactly the sequential one (crt_traverse.h: sample_radiance), so results do not
// depend on the schedule.  The kernel is bound by VALU issue in phase A and by the CU's one address unit (16 cycles per
// vector memory instruction whatever the number of lanes) in phase B, so the code below keeps
// scalar bookkeeping and register shuffling out of the loops, reads uniform data through scalar loads and the top of
// the tree from LDS; what was tried and rejected is listed in DESIGN.md section 7.
#ifndef CRT_WORK_STRIPES
#define CRT_WORK_STRIPES 64u // block counters: an atomic on ONE address costs the device ~5 ns, and every wave waits its turn
#endif
static_assert(CRT_WORK_STRIPES == 64u, "the order in which a wave visits the block counters is written for 64 of them");
#ifndef CRT_NODE_STEPS_PER_VOTE
#define CRT_NODE_STEPS_PER_VOTE 2
#endif
#ifndef CRT_NODE_CACHE_NODES
#define CRT_NODE_CACHE_NODES 236 // 4 waves x 4.5 KB of stack + 236 x 56 B of nodes = 31.6 KB per workgroup: five per CU = the same 20 waves (160 / 192 / 208 nodes in 64-byte records: 9.00 / 8.98 / 8.94 ms; 236 in 56-byte records: 8.88; at 32.1 KB only four workgroups fit a CU: 10.0 ms)
#endif
#ifndef CRT_LDS_VOTE_NUM
#define CRT_LDS_VOTE_NUM 3 // an LDS-only node step when at least NUM/DEN of the lanes at a node are at a cached node
#define CRT_LDS_VOTE_DEN 4
#endif
#ifndef CRT_LDS_STEPS_PER_VOTE
#define CRT_LDS_STEPS_PER_VOTE 4 // node steps per vote when the nodes come from the LDS copy
#endif
#ifndef CRT_WHOLE_TREE_NODES
#define CRT_WHOLE_TREE_NODES 1344 // 16 x 4.6 KB of stack + 84 KB of nodes = 156 KB: one workgroup per CU
#endif
#ifndef CRT_WAVES_PER_EU
#define CRT_WAVES_PER_EU 5 // 512 VGPRs / 5 waves -> 96 registers per lane; the few spills this forces sit on the rare tie path
#endif
// (a 16-wave workgroup is the whole-tree-in-LDS variant: one workgroup per CU = 4 waves per SIMD = 128 registers per lane)
#if CRT_WAVES_PER_EU > 0
#define CRT_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu((W == 16 && NC == CRT_WHOLE_TREE_NODES) ? 4 : CRT_WAVES_PER_EU, (W == 16 && NC == CRT_WHOLE_TREE_NODES) ? 4 : CRT_WAVES_PER_EU)))
#else
#define CRT_OCCUPANCY_ATTR
#endif
extern __shared__ int lds_dynamic[]; // render_kernel_dyn's stacks (DYNAMIC = 1) or node copy + stacks (DYNAMIC = 2), see there
// WHOLE: the launcher has checked that the whole node array is in the LDS copy (sc.nnodes <= NC): no device-memory node step exists
template <bool COUNT_WORK, int SD, bool SS, int W, int NC, bool RING, int DYNAMIC = 0, bool WHOLE = false, bool WIDE4 = false>
__device__ __forceinline__ void render_body(const SceneView &sc, const CameraPOD &cam, const RenderArgs &ra, float *__restrict__ out,
                                            int *__restrict__ ids, unsigned long long *__restrict__ counters)
{
    // W waves per workgroup, each on its own; with NC > 0 they share an LDS copy of the first NC nodes (breadth-first
    // numbering: the top of the tree).  The kernel is limited by the CU's address unit as much as by VALU issue (one more
    // dwordx4 load per node step costs 13 % of the frame), and about 70 % of the node visits of the headline scene go to the
    // first ~240 nodes: those are read from LDS.  W = 4, NC = 236, SD = 16: 4 x 4.5 KB of stack + 13 KB of nodes = 31.6 KB,
    // five workgroups per CU = the same 20 waves as with one wave per workgroup.
    float *lds_nodes;
    if constexpr (DYNAMIC == 2) lds_nodes = reinterpret_cast<float *>(lds_dynamic);
    else {
        __shared__ __align__(16) float lds_nodes_static[NC > 0 ? (WIDE4 ? CRT_LDS_WIDE_WORDS : CRT_LDS_NODE_WORDS) * NC : 1];
        lds_nodes = lds_nodes_static;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // SD = 14: trees at most 14 levels deep whose references fit 16 bits (launch_render checks both)
    typedef typename std::conditional<CRT_REG_TOS != 0, RegTopStack, typename std::conditional<SD == 14, LdsStack16, LdsStack>::type>::type stack_t;
    constexpr int NONE = stack_none<stack_t>::value;
    constexpr int STACK_INTS = (SD + 2) * 64 * (int)sizeof(typename stack_t::entry) / 4; // per wave
    int *lds_stack;
    if constexpr (DYNAMIC != 0) lds_stack = lds_dynamic + (DYNAMIC == 2 ? (WIDE4 ? CRT_LDS_WIDE_WORDS : CRT_LDS_NODE_WORDS) * NC : 0) + wave * STACK_INTS;
    else {
        __shared__ int lds_stack_all[W][STACK_INTS];
        lds_stack = lds_stack_all[wave];
    }
    CachedNodes node_src;
    node_src.global = sc.nodes; node_src.lds = (lds_q4_ptr)lds_nodes; node_src.ncached = NC > 0 ? min((int)sc.nnodes, NC) : 0;
    CachedNodes4 node_src4;
    node_src4.global = sc.nodes; node_src4.lds = (lds_q4_ptr)lds_nodes; node_src4.ncached = (uint32_t)node_src.ncached;
    if (NC > 0) {
        const float *g = reinterpret_cast<const float *>(sc.nodes);
        if constexpr (WIDE4) { // the first 28 words of the 32-word device record
            for (uint32_t k = threadIdx.x; k < CRT_LDS_WIDE_WORDS * (uint32_t)node_src.ncached; k += 64u * W) {
                const uint32_t node = k / CRT_LDS_WIDE_WORDS, j = k - node * CRT_LDS_WIDE_WORDS;
                lds_nodes[k] = g[32u * node + j];
            }
        } else {
            // word j of the 56-byte LDS record <- word of the 64-byte device record: the four rows' xyz, then the two references
            for (uint32_t k = threadIdx.x; k < CRT_LDS_NODE_WORDS * (uint32_t)node_src.ncached; k += 64u * W) {
                const uint32_t node = k / CRT_LDS_NODE_WORDS, j = k - node * CRT_LDS_NODE_WORDS;
                const uint32_t src = j < 12u ? (j / 3u) * 4u + j % 3u : (j == 12u ? 3u : 7u);
                lds_nodes[k] = g[16u * node + src];
            }
        }
        __syncthreads(); // the only workgroup barrier: from here on the waves do not meet again
    }
    // home counter of this wave: index mod 8 = workgroup mod 8 = the XCD the workgroup runs on (round robin)
    const uint32_t home = (blockIdx.x & 7u) | ((((blockIdx.x >> 3) * W + wave) & 7u) << 3);
    unsigned int *work = reinterpret_cast<unsigned int *>(counters + 32); // block counters, 64 dwords apart (crt_kernels.h)
    const uint32_t NO_PIXEL = 0xFFFFFFFFu;
    const float tmin = 0.001f; // camera.h:68

    stack_t stack;
    stack.init(reinterpret_cast<typename stack_t::entry *>(lds_stack) + lane);
    typename std::conditional<COUNT_WORK, RegCounters<SS>, RayCounters<SS>>::type cnt;
    cnt.refold_counter = counters + 4;

    uint32_t q = NO_PIXEL, ij = 0, pixel = 0; // ij = column | row << 16 of the pixel (both < 65536)
    f3 col = mk3(0.0f, 0.0f, 0.0f);
    int phase = P_FETCH;
    uint32_t rays_wave = 0;
    uint32_t blk_next = 0, blk_end = 0; // the wave's current block of work units: units [blk_next, blk_end) are still to hand out (wave-uniform)
    uint32_t blk_block = 0; // ... and its index in the frame
    // Sample chunks: the ring of blocks in flight (see finalize_ring_block).  ring_block[r] (LDS, touched once per block) = frame index
    // of the block in slot r; lane r of ring_count = units of it that are finished (a lane register read with v_readlane and written with a
    // select: an LDS round trip per shade phase was measurable).  Two scalar registers (the kernel has none to
    // spare): ring_state = slots without a block (bits 0..15) | slots whose block is complete and waits to be added up (16..31);
    // ring_aux = slot of the wave's current block (bits 0..3) | lanes that need a unit while the ring is full (8..14): those wait
    // in P_STARVED without counting as ready -- they would keep the wave in the shade phase while the walks that could complete a
    // block never advance -- and the others vote as a smaller wave.
    __shared__ uint32_t ring_block_all[W][16];
    // (the wave's index as a scalar and its row of ring_block_all are formed where they are needed -- once per block or per pass of
    // ring sums: kept across the main loop they were two of the five scalar registers the RING kernels spilled into lane registers)
#define CRT_WAVE_S ((uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))
#define CRT_RING_BLOCK (ring_block_all[CRT_WAVE_S])
    uint32_t ring_count = 0;
    uint32_t ring_state = RING ? (1u << ra.ring_slots) - 1u : 0u, ring_aux = 0;
    uint32_t stripe = home, stripes_left = CRT_WORK_STRIPES; // counter in use, counters not yet seen empty
    uint32_t samples = 0, depth = 0; // samples = index of the current sample | samples still to do << 16 (both <= 65535: crt_render's limits)
    Path p;
    p.o = p.d = p.attenuation = p.color = mk3(0.0f, 0.0f, 0.0f);
    Search hit;
    hit.t = FLT_MAX; hit.code = -1; hit.mat = 0; hit.rank = 0; hit.near_tie = false;
    Trav tv;
    tv.idir = tv.ood = mk3(0.0f, 0.0f, 0.0f); tv.shift = 0.0f; tv.hi = 0.0f; tv.cur = NONE;
    const int threshold_all = (int)ra.sched_threshold;
    uint32_t wc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; // schedule counters, lane 0 only (COUNT_WORK)
    unsigned long long t_start = 0;
    if constexpr (COUNT_WORK) t_start = wall_clock64(); // 100 MHz
    SectionClock<COUNT_WORK> clk; // [0] votes [1] hit + shade [2] store + fetch [3] Philox [4] scatter [5] camera [6] spheres + root box [7] node steps [8] leaf steps [9] adding up complete ring blocks
    clk.start();

    // Every pass of this loop makes progress (a shade phase retires or starts a path segment for every ready lane, a walk
    // phase advances at least one walk by a step), so it ends; the trip counter turns a scheduling bug into a flag the
    // host reports (counters[CRT_COUNTER_GUARD], checked by crt_render) instead of a wave that never leaves the GPU.
    bool guard_tripped = false;
    const uint32_t guard_limit = ra.guard_passes;
    for (uint32_t guard = 0;; guard++) {
        // sample chunks: complete blocks are added up and written out here, where few registers are live -- when ring_batch of them
        // have collected, or when the ring has no free slot left; their slots are free again and the lanes that waited may ask again
        if (RING && __builtin_expect((ring_aux & 0x80000000u) != 0u, 0)) { // (set in the shade phase when the ring's state says so)
            CRT_LAP(clk, 0);
            const RenderArgs rak = load_kernel_arg<RenderArgs>(offsetof(KernelArgsLayout, ra));
            const global_f32_ptr wring = (global_f32_ptr)(uintptr_t)((uint64_t)rak.ring_lo | (uint64_t)rak.ring_hi << 32) +
                                         (size_t)(rak.ring_wave_base + blockIdx.x * W + CRT_WAVE_S) * rak.ring_slots * rak.block_units * 3u;
            const uint32_t done = ring_state >> 16;
            finalize_ring_blocks(rak, out, wring, done, CRT_RING_BLOCK, lane);
            ring_count = (done >> lane) & 1u ? 0u : ring_count;
            ring_state = (ring_state & 0xFFFFu) | done; // done -> free
            ring_aux &= 0xFu;                            // nobody starved, nothing to add up
            phase = phase == P_STARVED ? P_FETCH : phase;
            CRT_LAP(clk, 9);
        }
        const bool walking = tv.cur != NONE;
        // (every lane is alive here, so masks of single comparisons can be combined as scalars; a ballot of `a && b` would
        // first write the bool to a register and compare it again)
        const unsigned long long walk_mask = __ballot(walking), active_mask = ~walk_mask & __ballot(phase > P_DONE);
        const int nwalk = popc64(walk_mask);
        const int nready = popc64(active_mask);
        if (nready + nwalk == 0) {
            // Starved lanes imply a full ring, i.e. units in flight on lanes that are ready or walking, so there are none left here;
            // if there ever were, the frame is incomplete and the host is told (below, behind the loop).  (No `continue` that would wake them instead: a
            // second back edge makes the compiler keep the loop's scalar state in lane registers -- 23 spills and half the speed.)
            break;
        }
        if (guard >= guard_limit) { guard_tripped = true; break; }
        const int threshold = RING ? max(threshold_all - (int)((ring_aux >> 8) & 0x7Fu), 1) : threshold_all;
        if (nready >= threshold || nwalk == 0) {
            // ================================ phase A ================================
            if constexpr (COUNT_WORK) { if (lane == 0) { wc[0]++; wc[1] += nready; } }
            CRT_LAP(clk, 0);
            const bool active = !walking && phase > P_DONE;
            // The shade phase reads the scene description (root box, sphere list, shading records, counts) from the kernel
            // arguments where it needs it, like the camera: kept in scalar registers for the whole kernel it does not fit
            // beside the walk's state and comes back through ~30 v_readlane per phase.  The walk itself only needs the
            // node and triangle pointers (`sc`).
            SceneView sck = load_kernel_arg<SceneView>(offsetof(KernelArgsLayout, sc));
            // (pointers rebuilt from raw argument words are generic to the compiler, i.e. flat_load: the node and triangle
            // pointers are the walk's own, and the shade phase reads spheres / shading records / triangles through
            // load_uniform / load_global)
            sck.nodes = sc.nodes; sck.tris = sc.tris;
            bool want_scatter = false;
            Surface sf;
            sf.point = sf.normal = mk3(0.0f, 0.0f, 0.0f); sf.front_face = false; sf.prim = -1;
            int m_type = MAT_DIFFUSE_LIGHT;
            float m_param = 0.0f;
            f3 colour = mk3(0.0f, 0.0f, 0.0f);
            if (active && phase == P_HIT) {
                float t = 0.0f;
                int mat = 0;
                const int code = resolve_hit_in_place<stack_t, decltype(cnt), WIDE4 ? 1 : 0>(sck, p.o, p.d, tmin, FLT_MAX, hit, t, mat, stack, cnt, tv); // (hit and tv are dead here)
                if (code < 0) { // camera.h:68-71
                    p.color = mk3(fma1(p.attenuation.x, cam.background.x, p.color.x), fma1(p.attenuation.y, cam.background.y, p.color.y),
                                  fma1(p.attenuation.z, cam.background.z, p.color.z));
                } else {
                    const Shade m = load_global(sck.shades, (size_t)mat); // issued together with the primitive fetch below: one memory round trip per hit
                    sf = surface_at<SS>(sck, code, p.o, p.d, t);
                    cnt.shade();
                    // (primary-hit ids: -1 was written when the unit that holds sample ids_sample was fetched; its first hit says otherwise)
                    if (depth == 0 && ra.sample_offset + (samples & 0xFFFFu) == ra.ids_sample && ids) ids[pixel] = sf.prim;
                    m_type = m.type; m_param = m.param;
                    colour = texture_value(m, sf.point);
                    const f3 e = emitted(m_type, colour); // camera.h:73-74
                    p.color = mk3(fma1(p.attenuation.x, e.x, p.color.x), fma1(p.attenuation.y, e.y, p.color.y), fma1(p.attenuation.z, e.z, p.color.z));
                    // camera.h:66,78: the scatter of the last allowed bounce cannot reach the image: not evaluated
                    want_scatter = scatters(m_type) && depth + 1 < ra.max_depth;
                }
                if (!want_scatter) { // path over: renderer.h:31
                    if constexpr (RING) phase = P_FETCH; // (one sample per unit: p.color goes to the ring)
                    else {
                        col = col + p.color;
                        samples += 1u - 0x10000u; // next index, one fewer to do
                        phase = samples >= 0x10000u ? P_NEW : P_FETCH;
                    }
                }
            }
            CRT_LAP(clk, 1);
            // ---- work queue, wave-uniform: finished units are stored, new ones come from the wave's current block.
            // Atomics on one address are served ~5 ns apart device-wide and every wave waits its turn (measured: 2 M per
            // frame cost 10 ms), hence blocks, and several counters.
            const bool fetching = active && phase == P_FETCH;
            const unsigned long long fm = __ballot(phase == P_FETCH); // (a lane that needs a unit is not walking; lanes that found the ring full ask again)
            if (fm) {
                const RenderArgs rak = load_kernel_arg<RenderArgs>(offsetof(KernelArgsLayout, ra)); // the frame geometry is only needed here
                constexpr bool ring_mode = RING; // (the launcher picks the kernel: RING <=> rak.chunks > 1)
                const uint32_t block_units = rak.block_units;
                if (ring_mode) {
                    // q = ring slot << 16 | index of the unit in its block; the unit's raw sum goes to the wave's ring
                    const global_f32_ptr wring = (global_f32_ptr)(uintptr_t)((uint64_t)rak.ring_lo | (uint64_t)rak.ring_hi << 32) +
                                                 (size_t)(rak.ring_wave_base + blockIdx.x * W + CRT_WAVE_S) * rak.ring_slots * block_units * 3u;
                    const bool done = fetching && q != NO_PIXEL;
                    if (done && pixel != NO_PIXEL) { // (pixel == NO_PIXEL: a unit outside the image, only counted)
                        const global_f32_ptr dst = wring + ((q >> 16) * block_units + (q & 0xFFFFu)) * 3u;
                        dst[0] = p.color.x; dst[1] = p.color.y; dst[2] = p.color.z;
                    }
                    // count the finished units per ring slot (they belong to one or two blocks, rarely more)
                    unsigned long long dm = __ballot(done);
                    while (dm) {
                        const uint32_t rs = (uint32_t)__builtin_amdgcn_readlane((int)(q >> 16), __builtin_ctzll(dm));
                        const unsigned long long m = dm & __ballot((q >> 16) == rs);
                        dm &= ~m;
                        const uint32_t count = (uint32_t)__builtin_amdgcn_readlane((int)ring_count, (int)rs) + (uint32_t)popc64(m);
                        ring_count = lane == rs ? count : ring_count;
                        if (count == block_units) ring_state |= 0x10000u << rs; // added up at the top of the main loop
                    }
                } else if (fetching && q != NO_PIXEL) store_unit(rak, out, q, col);
                // Units come in blocks of block_units (2^group_shift pixels x chunks) from CRT_WORK_STRIPES counters 256 B apart
                const uint32_t n_need = (uint32_t)popc64(fm), left = blk_end - blk_next, rk = (uint32_t)lane_rank(fm);
                uint32_t fresh_units = 0, fresh_block = 0, fresh_rs = 0;
                if (n_need > left && (!ring_mode || (ring_state & 0xFFFFu) != 0u)) {
                    const int leader = __builtin_ctzll(fm);
                    // blocks_per_stripe = 0: block b of counter k is global block b * stripes + k (all counters walk the frame
                    // together); else counter k owns the contiguous blocks [k * bps, (k + 1) * bps) (a band of the frame)
                    const uint32_t bps = rak.blocks_per_stripe, nblocks = rak.nblocks;
                    while (stripes_left > 0) {
                        // blocks of this counter
                        const uint32_t mine = bps ? (nblocks > stripe * bps ? (nblocks - stripe * bps < bps ? nblocks - stripe * bps : bps) : 0u)
                                                  : (nblocks + CRT_WORK_STRIPES - 1u - stripe) / CRT_WORK_STRIPES;
                        uint32_t b = 0;
                        if ((int)lane == leader) b = atomicAdd(work + stripe * 64u, 1u);
                        b = (uint32_t)__builtin_amdgcn_readlane((int)b, leader);
                        if (b < mine) {
                            fresh_block = bps ? stripe * bps + b : b * CRT_WORK_STRIPES + stripe;
                            fresh_units = block_units;
                            if (rak.chunk_mode == CM_WHOLE) { // (the last block of a whole-pixel frame may be short; with sample chunks a block is at most 64 pixels, and a rank's pixel slots are a multiple of 64)
                                const uint32_t behind = rak.local_pixels - fresh_block * block_units;
                                fresh_units = behind < block_units ? behind : block_units;
                            }
                            break;
                        }
                        // This counter is used up.  The wave that is the first to see that (b == mine) counts it in `used_up`;
                        // once all 64 are counted there is nothing left anywhere and a wave stops looking: without this every
                        // wave ends the frame by trying all 64 counters, one serialised atomic round trip each.
                        unsigned int *used_up = reinterpret_cast<unsigned int *>(counters + CRT_COUNTER_USED_UP);
                        if ((int)lane == leader && b == mine) atomicAdd(used_up, 1u);
                        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(used_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= CRT_WORK_STRIPES) {
                            stripes_left = 0;
                            break;
                        }
                        // else the next counter: first the ones with the same index mod 8 (workgroups go round robin over the 8
                        // XCDs, so those are the bands this XCD's L2 already serves), then the next residue
                        stripes_left--;
                        const uint32_t seen = CRT_WORK_STRIPES - stripes_left, s0 = home;
                        stripe = ((((s0 >> 3) + seen) & 7u) << 3) | (((s0 & 7u) + (seen >> 3)) & 7u);
                    }
                    if (ring_mode && fresh_units) { // the new block takes a free ring slot (its count is 0)
                        fresh_rs = (uint32_t)__builtin_ctz(ring_state);
                        ring_state &= ring_state - 1u;
                        CRT_RING_BLOCK[fresh_rs] = fresh_block;
                    }
                }
                // the lane's unit: index k in the wave's current block, or in the fresh one
                const bool from_current = rk < left;
                const uint32_t k = from_current ? blk_next + rk : rk - left;
                const bool got = from_current || k < fresh_units;
                const uint32_t block = from_current ? blk_block : fresh_block, rs = from_current ? ring_aux & 0xFu : fresh_rs;
                if (n_need > left) {
                    if (fresh_units) {
                        const uint32_t used = n_need - left;
                        blk_next = used < fresh_units ? used : fresh_units;
                        blk_end = fresh_units; blk_block = fresh_block;
                        ring_aux = (ring_aux & ~0xFu) | fresh_rs;
                    } else blk_next = blk_end;
                } else blk_next += n_need;
                if (ring_mode) { // complete blocks are added up (top of the loop) once ring_batch of them wait, or when no slot is free for a new block
                    const bool flush = (ring_state >> 16) != 0u && (popc64(ring_state >> 16) >= (int)ra.ring_batch || (ring_state & 0xFFFFu) == 0u);
                    ring_aux = flush ? ring_aux | 0x80000000u : ring_aux;
                }
                const bool exhausted = stripes_left == 0;
                if (ring_mode && !exhausted && n_need > left + fresh_units) {
                    ring_aux += (n_need - left - fresh_units) << 8; // lanes that will wait for a ring slot
                    if constexpr (COUNT_WORK) { if (lane == 0) wc[3] += n_need - left - fresh_units; } // (crt_stats.sched[3])
                }
                if (fetching) {
                    // Raster tile order on purpose.  Measured slower: fetching one pixel ahead (+15 %: a lane then holds a
                    // pixel hostage at the end of the frame), the tiles over the mesh first (+14 %), spread evenly (+5 %),
                    // or started at the mesh's first tile row (+11 %): DESIGN.md section 7.
                    q = NO_PIXEL;
                    if (!got) phase = exhausted ? P_DONE : ring_mode ? P_STARVED : P_FETCH; // (whole pixels: the block was short, asks again in the next phase A)
                    else {
                        // unit k of a block = part k % per_pixel of the block's pixel k / per_pixel: one sample of spp (ring), one chunk of
                        // n (slabs), the pixel itself (k < 65536, per_pixel < 65536: exact with ceil(2^32 / per_pixel))
                        const uint32_t g = rak.per_pixel > 1u ? __umulhi(k, rak.per_pixel_magic) : k, part = k - g * rak.per_pixel;
                        const uint32_t slot = (block << rak.group_shift) + g;
                        const bool inside = slot_pixel(rak, slot, ij);
                        if (inside) {
                            pixel = (ij >> 16) * rak.width + (ij & 0xFFFFu); // renderer.h:24
                            if constexpr (RING) {
                                samples = part;
                                if (ids && ra.sample_offset + samples == ra.ids_sample) ids[pixel] = -1; // until the sample's first hit says otherwise
                                q = (rs << 16) | k;
                            } else {
                                const uint32_t first = part * rak.chunk_spp;
                                samples = first | (rak.chunk_spp << 16);
                                col = mk3(0.0f, 0.0f, 0.0f);
                                if (ids && ra.ids_sample - (ra.sample_offset + first) < rak.chunk_spp) ids[pixel] = -1; // (unsigned: also false below the unit's first sample)
                                q = ((rak.flags & RA_COMPACT) ? slot : pixel) * rak.per_pixel + part;
                            }
                            phase = P_NEW;
                        } else if (ring_mode) { // outside the image: nothing to render, but the unit counts for its block (next phase A)
                            pixel = NO_PIXEL;
                            q = (rs << 16) | k;
                        } else if (rak.flags & RA_COMPACT) { // (a unit outside the image: its place in the compact list is cleared, the lane fetches again next time)
                            const size_t z = (size_t)slot * rak.per_pixel + part;
                            out[3 * z + 0] = 0.0f; out[3 * z + 1] = 0.0f; out[3 * z + 2] = 0.0f;
                        }
                    }
                }
            }
            CRT_LAP(clk, 2);
            bool has_ray = false;
            if (active) {
                // one Philox evaluation for both kinds of lane: stage 1 + depth for a scatter, stage 0 block 0 for a camera ray
                u4 r;
                r.x = r.y = r.z = r.w = 0u;
                // (the keys pass through an empty asm so that the ten round keys are scalar adds here, not loop invariants of the
                // whole kernel that live in spilled scalar registers and come back as v_readlane)
                uint32_t key0 = ra.key0, key1 = ra.key1;
                asm volatile("" : "+s"(key0), "+s"(key1));
                if (want_scatter || phase == P_NEW) r = philox4x32_10(pixel, ra.sample_offset + (samples & 0xFFFFu), want_scatter ? 1u + depth : 0u, 0u, key0, key1);
                CRT_LAP(clk, 3);
                if (want_scatter) {
                    f3 att, no, nd;
                    scatter(m_type, m_param, colour, p.d, sf.point, sf.normal, sf.front_face, r, att, no, nd);
                    p.attenuation = p.attenuation * att; // camera.h:82-83
                    p.o = no; p.d = nd;
                    depth++;
                    has_ray = true;
                }
                CRT_LAP(clk, 4);
                if (phase == P_NEW) {
                    const CameraPOD camk = load_kernel_arg<CameraPOD>(offsetof(KernelArgsLayout, cam));
                    camera_ray_from(camk, ij & 0xFFFFu, ij >> 16, pixel, ra.sample_offset + (samples & 0xFFFFu), key0, key1, r, p.o, p.d);
                    p.attenuation = mk3(1.0f, 1.0f, 1.0f);
                    p.color = mk3(0.0f, 0.0f, 0.0f);
                    depth = 0;
                    has_ray = true;
                }
                CRT_LAP(clk, 5);
                if (has_ray) {
                    hit.t = FLT_MAX; hit.code = -1; hit.near_tie = false;
                    trav_begin<MODE_CLOSEST>(sck, p.o, p.d, tmin, hit, 0, 0.0f, tv, stack, cnt);
                    phase = P_HIT; // what the lane does once tv.cur is CRT_BVH_NONE again (at once if the root box is missed)
                }
            }
            rays_wave += (uint32_t)popc64(__ballot(has_ray)); // path segments, counted per wave (uniform: a scalar register)
            CRT_LAP(clk, 6);
        } else {
            // ================================ phase B ================================
            const int alive = nready + nwalk;
            for (;;) {
                const bool inner = is_inner<NONE>(tv.cur), leaf = tv.cur < 0;
                const int n_inner = popc64(__ballot(inner)), n_leaf = popc64(__ballot(leaf));
                const int nw = n_inner + n_leaf;
                if (nw == 0 || alive - nw >= threshold) break;
                CRT_LAP(clk, 0);
                // With the LDS copy of the top of the tree there are two kinds of node step.  A vector memory instruction
                // occupies the CU's one address unit for 16 cycles whatever the number of active lanes, so a step that
                // fetches from device memory at all may as well do so for every lane at a node; when at least 3/4 of those
                // lanes are at cached nodes the step is LDS-only instead and the others wait.
                if constexpr (WIDE4) {
                    // four-child nodes: one kind of step, every lane at a node takes part and reads its node from the LDS copy or from
                    // device memory (the trees that get this format are far larger than any copy: most reads are device reads anyway)
                    if (n_inner >= n_leaf) {
                        if constexpr (COUNT_WORK) { if (lane == 0) { wc[4]++; wc[5] += n_inner; } }
                        if (inner) trav_inner4_from(node_src4, tv, stack, cnt);
#pragma unroll
                        for (int k = 1; k < CRT_NODE_STEPS_PER_VOTE; k++) {
                            const bool again = inner && is_inner<NONE>(tv.cur);
                            if constexpr (COUNT_WORK) { const int na = popc64(__ballot(again)); if (lane == 0) { wc[4]++; wc[5] += na; } }
                            if (again) trav_inner4_from(node_src4, tv, stack, cnt);
                        }
                        CRT_LAP(clk, 7);
                    } else {
                        if constexpr (COUNT_WORK) { if (lane == 0) { wc[6]++; wc[7] += n_leaf; } }
                        if (leaf) trav_leaf<MODE_CLOSEST>(sc, p.o, p.d, tmin, hit, 0, 0.0f, tv, stack, cnt);
                        CRT_LAP(clk, 8);
                    }
                    continue;
                }
                bool in_lds = false;
                int n_lds = 0;
                if constexpr (WHOLE) { in_lds = inner; n_lds = n_inner; }
                else if constexpr (NC > 0) {
                    in_lds = (unsigned)tv.cur < (unsigned)node_src.ncached; // leaf references and the end-of-walk value are beyond it as unsigned
                    n_lds = popc64(__ballot(in_lds));
                }
                if (NC > 0 && (WHOLE || n_lds * CRT_LDS_VOTE_DEN >= n_inner * CRT_LDS_VOTE_NUM) && n_lds >= n_leaf) {
                    if constexpr (COUNT_WORK) { if (lane == 0) { wc[4]++; wc[5] += n_lds; } }
                    const SignOffsets so = sign_offsets(tv);
                    if (in_lds) trav_inner_lds_sorted(node_src.lds, so, tv, stack, cnt);
#pragma unroll
                    for (int k = 1; k < CRT_LDS_STEPS_PER_VOTE; k++) {
                        const bool again = in_lds && (WHOLE ? is_inner<NONE>(tv.cur) : (unsigned)tv.cur < (unsigned)node_src.ncached);
                        if constexpr (COUNT_WORK) { const int na = popc64(__ballot(again)); if (lane == 0) { wc[4]++; wc[5] += na; } }
                        if (again) trav_inner_lds_sorted(node_src.lds, so, tv, stack, cnt);
                    }
                    CRT_LAP(clk, 7);
                } else if (!WHOLE && n_inner >= n_leaf) {
                    // (all lanes at a node take part: the loads cost the address unit the same whatever the number of lanes)
                    const bool mine = inner;
                    if constexpr (COUNT_WORK) { if (lane == 0) { wc[4]++; wc[5] += n_inner; } }
                    if (mine) trav_inner(sc, tv, stack, cnt);
                    // more node steps on the same vote: descents come in runs, and the vote costs half a step in scalar work
#pragma unroll
                    for (int k = 1; k < CRT_NODE_STEPS_PER_VOTE; k++) {
                        const bool again = mine && is_inner<NONE>(tv.cur);
                        if constexpr (COUNT_WORK) { const int na = popc64(__ballot(again)); if (lane == 0) { wc[4]++; wc[5] += na; } }
                        if (again) trav_inner(sc, tv, stack, cnt);
                    }
                    CRT_LAP(clk, 7);
                } else {
                    if constexpr (COUNT_WORK) { if (lane == 0) { wc[6]++; wc[7] += n_leaf; } }
                    if (leaf) trav_leaf<MODE_CLOSEST>(sc, p.o, p.d, tmin, hit, 0, 0.0f, tv, stack, cnt);
                    CRT_LAP(clk, 8);
                }
            }
        }
    }

    if (RING && (ring_state >> 16) != 0u) { // the complete blocks that were still collecting
        const RenderArgs rak = load_kernel_arg<RenderArgs>(offsetof(KernelArgsLayout, ra));
        const global_f32_ptr wring = (global_f32_ptr)(uintptr_t)((uint64_t)rak.ring_lo | (uint64_t)rak.ring_hi << 32) +
                                     (size_t)(rak.ring_wave_base + blockIdx.x * W + CRT_WAVE_S) * rak.ring_slots * rak.block_units * 3u;
        finalize_ring_blocks(rak, out, wring, ring_state >> 16, CRT_RING_BLOCK, lane);
    }
    if (RING && __ballot(phase == P_STARVED)) guard_tripped = true; // (see the loop's exit)
#undef CRT_RING_BLOCK
#undef CRT_WAVE_S
    uint32_t refolds = 0;
    if constexpr (COUNT_WORK) refolds = wave_sum(cnt.refolds);
    if (lane == 0) {
        atomicAdd(&counters[0], (unsigned long long)rays_wave);
        if (refolds) atomicAdd(&counters[4], (unsigned long long)refolds);
        if (guard_tripped) atomicAdd(&counters[CRT_COUNTER_GUARD], 1ull); // must stay 0
    }
    if constexpr (COUNT_WORK) {
        const uint32_t n = wave_sum(cnt.nodes), t = wave_sum(cnt.tris), sp = wave_sum(cnt.spheres), sh = wave_sum(cnt.shades);
        if (lane == 0) {
            atomicAdd(&counters[5], (unsigned long long)sh);
            atomicAdd(&counters[1], (unsigned long long)n);
            atomicAdd(&counters[2], (unsigned long long)t);
            atomicAdd(&counters[3], (unsigned long long)sp);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t w = wave_sum(wc[k]);
            if (lane == 0) atomicAdd(&counters[8 + k], (unsigned long long)w);
        }
        // load balance: waves that did work, the sum and the maximum of their lifetimes (10 ns ticks)
        if (lane == 0 && wc[0] > 1) {
            const unsigned long long dt = wall_clock64() - t_start;
            atomicAdd(&counters[16], 1ull);
            atomicAdd(&counters[17], dt);
            atomicMax(&counters[10], dt);
#pragma unroll
            for (int k = 0; k < 10; k++) atomicAdd(&counters[18 + k], clk.sum[k]); // section clocks (CRT_PRINT_SECTIONS)
        }
    }
}

template <bool COUNT_WORK, int SD, bool SS, int W, int NC, bool RING, bool WIDE4 = false>
__global__ __launch_bounds__(64 * W) CRT_OCCUPANCY_ATTR void render_kernel(SceneView sc, CameraPOD cam, RenderArgs ra, float *__restrict__ out,
                                                                       int *__restrict__ ids, unsigned long long *__restrict__ counters)
{
    render_body<COUNT_WORK, SD, SS, W, NC, RING, 0, false, WIDE4>(sc, cam, ra, out, ids, counters);
}

// The 16-wave workgroup of a two-launch frame (launch_split).  With all of its LDS declared (one workgroup per CU) the compiler
// assumes four waves per SIMD and hands out up to 128 registers whatever the attributes say; the kernel takes 99-101, is allocated
// 104, and beside the small workgroup's 96 that is all 512 registers of a SIMD lane: nothing else gets onto the CU while a frame
// is resident, and the 21-register finalize pass of the PREVIOUS frame -- which the next frame on its stream waits for -- only
// runs when the frame drains.  So the stacks (addressed through a per-lane pointer anyway) are dynamic LDS, whose size the
// compiler does not see (DYNAMIC = 1), the five-waves-per-SIMD attribute holds the kernel to 96 registers, and the pair leaves
// the same 32 registers per SIMD lane free as five 96-register waves do.  The small workgroup of the pair has the same problem
// from 32 KB of LDS on (four waves per SIMD by LDS -> 128 registers): with node copy AND stacks dynamic (DYNAMIC = 2) it can use
// whatever the large one leaves.
template <int SD, int W, int NC, int DYNAMIC, bool WIDE4 = false> constexpr size_t dynamic_lds_bytes()
{
    return (size_t)W * (SD + 2) * 64 * (SD == 14 ? sizeof(short) : sizeof(int)) + // (14 levels: 16-bit entries)
           (DYNAMIC == 2 ? (size_t)(WIDE4 ? 4u * CRT_LDS_WIDE_WORDS : CRT_LDS_NODE_BYTES) * NC : 0);
}
template <bool COUNT_WORK, int SD, bool SS, int W, int NC, bool RING, int DYNAMIC, bool WHOLE = false, bool WIDE4 = false>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(CRT_WAVES_PER_EU, CRT_WAVES_PER_EU)))
void render_kernel_dyn(SceneView sc, CameraPOD cam, RenderArgs ra, float *__restrict__ out, int *__restrict__ ids, unsigned long long *__restrict__ counters)
{
    render_body<COUNT_WORK, SD, SS, W, NC, RING, DYNAMIC, WHOLE, WIDE4>(sc, cam, ra, out, ids, counters);
}

#ifdef CRT_WITH_POOL_KERNEL // experimental, measured and rejected: built with EXTRA_DEVFLAGS=-DCRT_WITH_POOL_KERNEL only
// =========================================================================================================
// render_pool_kernel: the same per-path operation sequence, scheduled over a pool of pixel contexts in LDS.
//
// render_kernel above binds one pixel to one lane, so a phase only ever fills the lanes whose pixel happens to be in
// that phase (32 % of the lanes of a walk step, 37 % of the lanes in the shading code on the headline scene) and the
// kernel is bound by VALU issue.  Here a wave owns P > 64 pixel contexts whose state lives in LDS (structure of arrays,
// one dword per field and context), and three LDS lists say what each context needs next:
//     cam   : a camera ray (after a path ended; also "needs a pixel")
//     shade : a surface hit to shade and scatter
//     walk  : a ray that entered the triangle BVH and waits for a lane
// Lanes are not bound to contexts.  The wave runs, by priority: walk steps while at least 64-REFILL lanes are walking;
// else it retires finished walks (their hit goes back to the context; a tie is resolved here, on the lane's own stack)
// and refills free lanes from the walk list; else a batch of up to 64 contexts of the longer of the shade / cam lists.
// Every batch is compact, so the shading code, the camera code and the walk steps run with (nearly) full waves.
// A context still performs exactly the sequential operation sequence of its pixel (samples in order), so the image is
// bit-identical to render_kernel's.
// =========================================================================================================
enum { CF_IJ = 0, CF_PACK, CF_OX, CF_OY, CF_OZ, CF_DX, CF_DY, CF_DZ, CF_AX, CF_AY, CF_AZ, CF_CX, CF_CY, CF_CZ, CF_T, CF_CODE, CF_MAT, CF_COUNT };
// CF_IJ: pixel column (bits 0..15) | row (16..31); 0xFFFFFFFF = no pixel.
// CF_PACK: sample index (bits 0..15) | depth (16..23) | tag (24..31)
enum { TAG_FETCH = 0, TAG_MISS = 1, TAG_END = 2, TAG_RUN = 3 };
enum { DEST_NONE = 0, DEST_WALK = 1, DEST_SHADE = 2, DEST_CAM = 3 };
#define CRT_TIE_BIT 0x40000000


template <int P>
struct Pool {
    uint32_t *f; // [CF_COUNT][P]
    __device__ __forceinline__ uint32_t u(int field, int c) const { return f[field * P + c]; }
    __device__ __forceinline__ int i(int field, int c) const { return (int)f[field * P + c]; }
    __device__ __forceinline__ float fl(int field, int c) const { return __uint_as_float(f[field * P + c]); }
    __device__ __forceinline__ f3 v3(int field, int c) const { return mk3(fl(field, c), fl(field + 1, c), fl(field + 2, c)); }
    __device__ __forceinline__ void set(int field, int c, uint32_t v) { f[field * P + c] = v; }
    __device__ __forceinline__ void set(int field, int c, int v) { f[field * P + c] = (uint32_t)v; }
    __device__ __forceinline__ void set(int field, int c, float v) { f[field * P + c] = __float_as_uint(v); }
    __device__ __forceinline__ void set3(int field, int c, f3 v) { set(field, c, v.x); set(field + 1, c, v.y); set(field + 2, c, v.z); }
};

// wave-uniform list push: call in uniform control flow only (n stays in a scalar register)
typedef unsigned short ctx_id;
__device__ __forceinline__ void list_push(ctx_id *list, int &n, bool pred, int c)
{
    const unsigned long long m = __ballot(pred);
    if (pred) list[n + lane_rank(m)] = (ctx_id)c;
    n += popc64(m);
}

// Start of a path segment for context c (camera.h:68 world->hit): the sphere tests and the root-box test.  Writes the
// ray and the search state to the context and says where the context goes next.
template <int P, class Counters>
__device__ __forceinline__ int ray_start(const SceneView &sc, Pool<P> &pool, int c, f3 o, f3 d, float tmin, uint32_t pack_run, Counters &cnt)
{
    cnt.ray();
    Search s;
    s.t = FLT_MAX; s.code = -1; s.mat = 0; s.rank = 0; s.near_tie = false;
    trav_spheres<MODE_CLOSEST>(sc, o, d, tmin, s, 0, 0.0f, cnt);
    Trav tv;
    trav_root<MODE_CLOSEST>(sc, o, d, s, 0.0f, tv);
    pool.set3(CF_OX, c, o);
    pool.set3(CF_DX, c, d);
    if (tv.cur != CRT_BVH_NONE || s.near_tie) { // a tie among spheres is resolved where walks retire (needs a stack)
        pool.set(CF_T, c, s.t); pool.set(CF_CODE, c, s.code); pool.set(CF_MAT, c, s.mat | (s.near_tie ? CRT_TIE_BIT : 0));
        pool.set(CF_PACK, c, pack_run);
        return DEST_WALK;
    }
    if (s.code < 0) { // camera.h:68-71: nothing hit
        pool.set(CF_PACK, c, (pack_run & 0x00FFFFFFu) | ((uint32_t)TAG_MISS << 24));
        return DEST_CAM;
    }
    pool.set(CF_T, c, s.t); pool.set(CF_CODE, c, s.code); pool.set(CF_MAT, c, s.mat);
    pool.set(CF_PACK, c, pack_run);
    return DEST_SHADE;
}

// LDS traffic between lanes of one wave needs no hardware barrier (a wave's LDS operations complete in order); this
// only keeps the compiler from moving them across the hand-over points.
__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// W waves per workgroup, each with its own pool, lists and stack; they share the LDS copy of the first NC nodes and never
// synchronise after it is filled.
template <bool COUNT_WORK, int P, int SD, int W, int NC>
__global__ __launch_bounds__(64 * W) void render_pool_kernel(SceneView sc, CameraPOD cam, RenderArgs ra, float *__restrict__ out,
                                                             int *__restrict__ ids, unsigned long long *__restrict__ counters)
{
    __shared__ q4 lds_nodes[NC > 0 ? 4 * NC : 1];
    __shared__ int lds_stack_all[W][(SD + 2) * 64];
    __shared__ uint32_t lds_pool_all[W][CF_COUNT * P];
    __shared__ ctx_id l_cam_all[W][P], l_shade_all[W][P], l_walk_all[W][P];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    int *lds_stack = lds_stack_all[wave];
    uint32_t *lds_pool = lds_pool_all[wave];
    ctx_id *l_cam = l_cam_all[wave], *l_shade = l_shade_all[wave], *l_walk = l_walk_all[wave];
    CachedNodes nodes;
    nodes.global = sc.nodes; nodes.lds = (lds_q4_ptr)lds_nodes; nodes.ncached = NC > 0 ? min((int)sc.nnodes, NC) : 0;
    if (NC > 0) {
        for (int k = (int)threadIdx.x; k < 4 * nodes.ncached; k += 64 * W) lds_nodes[k] = sc.nodes[k];
        __syncthreads(); // the only workgroup barrier
    }
    unsigned int *work = reinterpret_cast<unsigned int *>(counters + 6);
    const uint32_t NO_PIXEL = 0xFFFFFFFFu;
    const float tmin = 0.001f; // camera.h:68
    const int REFILL = (int)ra.sched_threshold; // free lanes that trigger a refill / end a run of walk steps

    LdsStack stack;
    stack.init(lds_stack + lane);
    Pool<P> pool;
    pool.f = lds_pool;
    typename std::conditional<COUNT_WORK, RegCounters<true>, RayCounters<true>>::type cnt;
    cnt.refold_counter = counters + 4;
    uint32_t wc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long t_start = 0;
    if constexpr (COUNT_WORK) t_start = wall_clock64();

    // all contexts start by asking for a pixel
    for (int c = (int)lane; c < P; c += 64) {
        pool.set(CF_IJ, c, NO_PIXEL);
        pool.set(CF_PACK, c, (uint32_t)TAG_FETCH << 24);
        l_cam[c] = (ctx_id)c;
    }
    int n_cam = P, n_shade = 0, n_walkq = 0, live = P;
    wave_fence();

    // walking state of this lane
    int my = -1;
    f3 o = mk3(0.0f, 0.0f, 0.0f), d = o;
    Search hit;
    hit.t = FLT_MAX; hit.code = -1; hit.mat = 0; hit.rank = 0; hit.near_tie = false;
    Trav tv;
    tv.idir = tv.ood = mk3(0.0f, 0.0f, 0.0f); tv.shift = 0.0f; tv.hi = 0.0f; tv.cur = CRT_BVH_NONE;

    for (uint32_t guard = 0; guard < 50000000u; guard++) { // every wave leaves this loop: all contexts retire, or the guard trips
        const int n_walk = popc64(__ballot(tv.cur != CRT_BVH_NONE));
        const bool others = n_walkq > 0 || n_shade > 0 || n_cam > 0;
        if (n_walk > 0 && (n_walk >= 64 - REFILL || !others)) {
            // ================================ walk steps ================================
            // until REFILL lanes are free (then they are retired and refilled, or their hits are shaded)
            const int low = max((others ? 64 : n_walk) - REFILL + (others ? 0 : 1), 1);
            for (;;) {
                const bool inner = is_inner(tv.cur), leaf = tv.cur < 0;
                const int n_inner = popc64(__ballot(inner)), n_leaf = popc64(__ballot(leaf));
                if (n_inner + n_leaf < low) break;
                if (n_inner >= n_leaf) {
                    if constexpr (COUNT_WORK) { if (lane == 0) { wc[4]++; wc[5] += n_inner; } }
                    if (inner) trav_inner_from(nodes, tv, stack, cnt);
                } else {
                    if constexpr (COUNT_WORK) { if (lane == 0) { wc[6]++; wc[7] += n_leaf; } }
                    if (leaf) trav_leaf<MODE_CLOSEST>(sc, o, d, tmin, hit, 0, 0.0f, tv, stack, cnt);
                }
            }
            continue;
        }
        // ================================ retire finished walks ================================
        {
            const bool fin = my >= 0 && tv.cur == CRT_BVH_NONE;
            int dest = DEST_NONE;
            const int c = my;
            if (fin) {
                float t = 0.0f;
                int mat = 0;
                const int code = resolve_hit(sc, o, d, tmin, FLT_MAX, hit, t, mat, stack, cnt); // tie fold (rare) on this lane's stack
                if (code < 0) {
                    pool.set(CF_PACK, c, (pool.u(CF_PACK, c) & 0x00FFFFFFu) | ((uint32_t)TAG_MISS << 24));
                    dest = DEST_CAM;
                } else {
                    pool.set(CF_T, c, t); pool.set(CF_CODE, c, code); pool.set(CF_MAT, c, mat);
                    dest = DEST_SHADE;
                }
                my = -1;
            }
            list_push(l_shade, n_shade, dest == DEST_SHADE, c);
            list_push(l_cam, n_cam, dest == DEST_CAM, c);
        }
        if (n_walkq > 0) {
            // ================================ refill free lanes from the walk list ================================
            const bool is_free = my < 0;
            const unsigned long long m = __ballot(is_free);
            const int nt = min(popc64(m), n_walkq);
            const int r = lane_rank(m);
            if (is_free && r < nt) {
                const int c = l_walk[n_walkq - 1 - r];
                my = c;
                o = pool.v3(CF_OX, c);
                d = pool.v3(CF_DX, c);
                const int mt = pool.i(CF_MAT, c);
                hit.t = pool.fl(CF_T, c); hit.code = pool.i(CF_CODE, c); hit.mat = mt & ~CRT_TIE_BIT; hit.near_tie = (mt & CRT_TIE_BIT) != 0;
                trav_root<MODE_CLOSEST>(sc, o, d, hit, 0.0f, tv);
                stack.reset();
            }
            n_walkq -= nt;
            wave_fence();
            continue;
        }
        if (n_shade > 0 && n_shade >= n_cam) {
            // ================================ shade batch ================================
            const int nb = min(n_shade, 64);
            const bool have = (int)lane < nb;
            const int c = have ? l_shade[n_shade - nb + (int)lane] : 0;
            n_shade -= nb;
            if constexpr (COUNT_WORK) { if (lane == 0) { wc[0]++; wc[1] += nb; } }
            int dest = DEST_NONE;
            if (have) {
                const f3 ro = pool.v3(CF_OX, c), rd = pool.v3(CF_DX, c);
                const float t = pool.fl(CF_T, c);
                const int code = pool.i(CF_CODE, c), mat = pool.i(CF_MAT, c);
                const uint32_t pack = pool.u(CF_PACK, c);
                const uint32_t s_idx = pack & 0xFFFFu, depth = (pack >> 16) & 0xFFu;
                const f3 att = pool.v3(CF_AX, c);
                const Shade m = sc.shades[mat]; // issued together with the primitive fetch: one memory round trip
                const Surface sf = surface_at(sc, code, ro, rd, t);
                cnt.shade();
                const uint32_t ij = pool.u(CF_IJ, c);
                const uint32_t pixel = (ij >> 16) * ra.width + (ij & 0xFFFFu); // renderer.h:24
                if (depth == 0 && ids && ra.sample_offset + s_idx == ra.ids_sample) ids[pixel] = sf.prim;
                const f3 colour = texture_value(m, sf.point);
                if (scatters(m.type) && depth + 1 < ra.max_depth) {
                    const u4 r = philox4x32_10(pixel, ra.sample_offset + s_idx, 1u + depth, 0u, ra.key0, ra.key1);
                    f3 a2, no, nd;
                    scatter(m.type, m.param, colour, rd, sf.point, sf.normal, sf.front_face, r, a2, no, nd);
                    pool.set3(CF_AX, c, att * a2); // camera.h:82-83
                    dest = ray_start(sc, pool, c, no, nd, tmin, s_idx | ((depth + 1u) << 16) | ((uint32_t)TAG_RUN << 24), cnt);
                } else {
                    // path over (camera.h:73-79): its colour is the emission seen through the attenuation; a path only
                    // ever collects one non-zero term (lights do not scatter, material.h:14-16), so colour = fma(att, e, 0)
                    const f3 e = emitted(m.type, colour);
                    pool.set3(CF_AX, c, mk3(fma1(att.x, e.x, 0.0f), fma1(att.y, e.y, 0.0f), fma1(att.z, e.z, 0.0f)));
                    pool.set(CF_PACK, c, (pack & 0x00FFFFFFu) | ((uint32_t)TAG_END << 24));
                    dest = DEST_CAM;
                }
            }
            wave_fence();
            list_push(l_walk, n_walkq, dest == DEST_WALK, c);
            list_push(l_shade, n_shade, dest == DEST_SHADE, c);
            list_push(l_cam, n_cam, dest == DEST_CAM, c);
            wave_fence();
            continue;
        }
        if (n_cam > 0) {
            // ================================ camera batch ================================
            const int nb = min(n_cam, 64);
            const bool have = (int)lane < nb;
            const int c = have ? l_cam[n_cam - nb + (int)lane] : 0;
            n_cam -= nb;
            if constexpr (COUNT_WORK) { if (lane == 0) { wc[0]++; wc[1] += nb; wc[3]++; } }
            uint32_t ij = NO_PIXEL, s_idx = 0;
            bool need_pixel = false;
            if (have) {
                const uint32_t pack = pool.u(CF_PACK, c);
                const uint32_t tag = pack >> 24;
                s_idx = pack & 0xFFFFu;
                ij = pool.u(CF_IJ, c);
                if (tag != TAG_FETCH) { // a path ended: renderer.h:31
                    const f3 att = pool.v3(CF_AX, c);
                    const f3 color = tag == TAG_MISS ? mk3(fma1(att.x, cam.background.x, 0.0f), fma1(att.y, cam.background.y, 0.0f),
                                                           fma1(att.z, cam.background.z, 0.0f))
                                                     : att; // TAG_END: the path's colour was left in the attenuation fields
                    pool.set3(CF_CX, c, pool.v3(CF_CX, c) + color);
                    s_idx++;
                }
                need_pixel = tag == TAG_FETCH || s_idx >= ra.spp;
                if (need_pixel && ij != NO_PIXEL) { // pixel complete
                    const uint32_t i = ij & 0xFFFFu, j = ij >> 16;
                    const uint32_t tile = (j >> 3) * ra.tiles_x + (i >> 3);
                    store_unit(ra, out, (ra.flags & RA_COMPACT) ? 64u * (tile / ra.tile_ranks) + ((j & 7u) << 3 | (i & 7u)) : j * ra.width + i, pool.v3(CF_CX, c));
                }
            }
            // pixel queue: one atomic per batch
            const unsigned long long fm = __ballot(need_pixel);
            bool retired = false, retry = false;
            if (fm) {
                unsigned int base = 0;
                if (lane == (uint32_t)__builtin_ctzll(fm)) base = atomicAdd(work, (unsigned int)popc64(fm));
                base = (unsigned int)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(fm));
                if (need_pixel) {
                    const uint32_t q = base + (uint32_t)lane_rank(fm);
                    if (q >= ra.local_pixels) { // queue empty: the context retires
                        retired = true;
                    } else {
                        const uint32_t tile = (q >> 6) * ra.tile_ranks + ra.tile_rank, l = q & 63u;
                        const uint32_t i = (tile % ra.tiles_x) * 8u + (l & 7u), j = (tile / ra.tiles_x) * 8u + (l >> 3);
                        if (i < ra.width && j < ra.height) { // renderer.h:22
                            s_idx = 0;
                            ij = i | (j << 16);
                            pool.set(CF_IJ, c, ij);
                            pool.set3(CF_CX, c, mk3(0.0f, 0.0f, 0.0f));
                        } else { // outside the image (ragged last tile column / row): nothing to render, ask again
                            if (ra.flags & RA_COMPACT) { out[3 * (size_t)q + 0] = 0.0f; out[3 * (size_t)q + 1] = 0.0f; out[3 * (size_t)q + 2] = 0.0f; }
                            pool.set(CF_IJ, c, NO_PIXEL);
                            pool.set(CF_PACK, c, (uint32_t)TAG_FETCH << 24);
                            retry = true;
                        }
                    }
                }
            }
            live -= popc64(__ballot(retired));
            int dest = DEST_NONE;
            if (have && !retired) {
                if (retry) dest = DEST_CAM;
                else {
                    const uint32_t i = ij & 0xFFFFu, j = ij >> 16, pixel = j * ra.width + i;
                    const uint32_t sample = ra.sample_offset + s_idx;
                    if (ids && sample == ra.ids_sample) ids[pixel] = -1; // until this sample's first hit says otherwise (shade batch)
                    f3 co, cd;
                    camera_ray(cam, i, j, pixel, sample, ra.key0, ra.key1, co, cd);
                    pool.set3(CF_AX, c, mk3(1.0f, 1.0f, 1.0f));
                    dest = ray_start(sc, pool, c, co, cd, tmin, s_idx | ((uint32_t)TAG_RUN << 24), cnt);
                }
            }
            wave_fence();
            list_push(l_walk, n_walkq, dest == DEST_WALK, c);
            list_push(l_shade, n_shade, dest == DEST_SHADE, c);
            list_push(l_cam, n_cam, dest == DEST_CAM, c);
            wave_fence();
            continue;
        }
        if (n_walk == 0 && popc64(__ballot(my >= 0)) == 0) break; // nothing listed, nothing walking, nothing to retire
    }

    const uint32_t rays = wave_sum(cnt.rays);
    uint32_t refolds = 0;
    if constexpr (COUNT_WORK) refolds = wave_sum(cnt.refolds);
    if (lane == 0) {
        atomicAdd(&counters[0], (unsigned long long)rays);
        if (refolds) atomicAdd(&counters[4], (unsigned long long)refolds);
        if (live != 0) atomicAdd(&counters[CRT_COUNTER_GUARD], 1ull); // must stay 0: a wave left with unfinished contexts (guard tripped)
    }
    if constexpr (COUNT_WORK) {
        const uint32_t n = wave_sum(cnt.nodes), t = wave_sum(cnt.tris), sp = wave_sum(cnt.spheres), sh = wave_sum(cnt.shades);
        if (lane == 0) {
            atomicAdd(&counters[5], (unsigned long long)sh);
            atomicAdd(&counters[1], (unsigned long long)n);
            atomicAdd(&counters[2], (unsigned long long)t);
            atomicAdd(&counters[3], (unsigned long long)sp);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t w = wave_sum(wc[k]);
            if (lane == 0) atomicAdd(&counters[8 + k], (unsigned long long)w);
        }
        if (lane == 0 && wc[0] > 1) {
            const unsigned long long dt = wall_clock64() - t_start;
            atomicAdd(&counters[16], 1ull);
            atomicAdd(&counters[17], dt);
            atomicMax(&counters[10], dt);
        }
    }
}

#endif // CRT_WITH_POOL_KERNEL

// compact (rank-major, tile-major) -> row-major frame
__global__ __launch_bounds__(64) void untile_kernel(const float *__restrict__ compact_all, uint32_t width, uint32_t height, uint32_t tiles_x,
                                                    uint32_t ntiles, uint32_t tile_ranks, uint32_t tiles_per_rank, float *__restrict__ out)
{
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    if (tile >= ntiles) return;
    const uint32_t rank = tile % tile_ranks, slot = tile / tile_ranks;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t i = tx * 8u + (lane & 7u), j = ty * 8u + (lane >> 3);
    if (i >= width || j >= height) return;
    const size_t src = ((size_t)rank * tiles_per_rank + slot) * 64u + lane;
    const size_t dst = (size_t)j * width + i;
    out[3 * dst + 0] = compact_all[3 * src + 0];
    out[3 * dst + 1] = compact_all[3 * src + 1];
    out[3 * dst + 2] = compact_all[3 * src + 2];
}

// accum / total_spp, then gamma (renderer.h:36-37, color.h:5-24)
__global__ __launch_bounds__(256) void finalize_kernel(const float *__restrict__ accum, size_t n, float spp, float *__restrict__ out)
{
    const size_t k = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const float v = accum[k] / spp;
    out[k] = v > 0 ? __builtin_sqrtf(v) : 0.0f;
}

// sample-split frames of several ranks on one device: out[k] = finish(((frame 0 + frame 1) + frame 2) + ...), frames n floats
// apart, finish = identity (raw sums) or /spp and gamma
__global__ __launch_bounds__(256) void sum_frames_kernel(const float *__restrict__ frames, uint32_t count, size_t n, float spp, int raw,
                                                         float *__restrict__ out)
{
    const size_t k = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    float v = frames[k];
    for (uint32_t c = 1; c < count; c++) v = v + frames[(size_t)c * n + k];
    if (!raw) {
        v = v / spp;
        v = v > 0 ? __builtin_sqrtf(v) : 0.0f;
    }
    out[k] = v;
}

// Sample chunks through slabs (RenderArgs.chunk_mode CM_SLAB): `slabs` is [pixel][chunk][3], the raw chunk sums of a pixel side by
// side (the lanes that render them sit in one wave: with 64 chunks their twelve-byte stores fill 768 contiguous bytes).
// out[pixel][ch] = finish((((0 + chunk 0) + chunk 1) + ...): the sum starts at 0 like the reference's pixel_color (renderer.h:26), so
// with one sample per chunk the float operations are the sequential loop's own.  21 registers on purpose: the pass runs beside the
// next frame's render kernel, in the 32 registers per SIMD lane that five 96-register waves leave free.
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(32))) void finalize_chunks_kernel(const float *__restrict__ slabs, uint32_t chunks, size_t npixels, float spp, int raw,
                                                             float *__restrict__ out)
{
    const size_t pixel = (size_t)blockIdx.x * 64u + threadIdx.x;
    if (pixel >= npixels) return;
    const float *row = slabs + pixel * chunks * 3u;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if ((chunks & 3u) == 0u) { // four chunks = 48 bytes = three 16-byte loads (the row starts at a multiple of 48 bytes)
        const v4f *q = reinterpret_cast<const v4f *>(row);
#pragma unroll 1
        for (uint32_t c = 0; c < chunks; c += 4u, q += 3) {
            const v4f x = q[0], y = q[1], z = q[2];
            r = r + x.x; g = g + x.y; b = b + x.z;
            r = r + x.w; g = g + y.x; b = b + y.y;
            r = r + y.z; g = g + y.w; b = b + z.x;
            r = r + z.y; g = g + z.z; b = b + z.w;
        }
    } else {
        for (uint32_t c = 0; c < chunks; c++) { r = r + row[3u * c]; g = g + row[3u * c + 1u]; b = b + row[3u * c + 2u]; }
    }
    out[3u * pixel + 0u] = finish_channel(r, spp, raw != 0);
    out[3u * pixel + 1u] = finish_channel(g, spp, raw != 0);
    out[3u * pixel + 2u] = finish_channel(b, spp, raw != 0);
}

hipError_t launch_finalize_chunks(const float *slabs, uint32_t chunks, size_t n, uint32_t spp, bool raw, float *out, hipStream_t stream)
{
    const size_t npixels = n / 3;
    const uint32_t blocks = (uint32_t)((npixels + 63) / 64);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(finalize_chunks_kernel, dim3(blocks), dim3(64), 0, stream, slabs, chunks, npixels, (float)spp, raw ? 1 : 0, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Static scratch of crt_render_noalloc (include/crt.h): counters, then the frame's chunk slabs [pixel][chunk][3] or the waves' rings
// (crt_kernels.h: CM_SLAB / CM_RING), for callers that may not allocate.
// `scratch` is a host-side stand-in for the symbol, not a device address: host code only ever uses the per-device
// address hipGetSymbolAddress returned, taken once per device with that device current (each GPU holds its own copy of
// the array at its own address) and handed to the kernels as an argument like any other buffer.
// ---------------------------------------------------------------------------------------------------------
static_assert(CRT_COUNTER_BYTES <= CRT_NOALLOC_SLAB_OFFSET && CRT_NOALLOC_SLAB_OFFSET % 256u == 0, "counter block must fit in front of the slabs");
static __device__ __align__(256) unsigned char scratch[CRT_NOALLOC_SLAB_OFFSET + CRT_NOALLOC_SLAB_BYTES];
namespace {
constexpr int kMaxDevices = 64;
void *g_scratch_ptr[kMaxDevices];
std::mutex g_scratch_mutex;
}

hipError_t noalloc_scratch_init(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return hipSuccess;
    int current = -1;
    hipError_t e = hipGetDevice(&current);
    if (e != hipSuccess) return e;
    if (current != device) return hipErrorInvalidDevice; // the symbol's address is per device: the caller makes `device` current first
    void *ptr = nullptr;
    e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

void *noalloc_scratch_ptr(int device)
{
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    return g_scratch_ptr[device];
}

hipError_t launch_sum_frames(const float *frames, uint32_t count, size_t n, uint32_t spp, bool raw, float *out, hipStream_t stream)
{
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(sum_frames_kernel, dim3(blocks), dim3(256), 0, stream, frames, count, n, (float)spp, raw ? 1 : 0, out);
    return hipGetLastError();
}

// node_cache = 3: one 16-wave workgroup per CU holding the WHOLE node array in LDS (trees of at most CRT_WHOLE_TREE_NODES
// nodes and 16 levels): no node is ever fetched from device memory.  Only instantiated for the 16-level stack (24 / 32
// levels would not fit 160 KB).
template <bool CW, int SD, bool SS, bool RING>
static bool launch_whole_tree(const SceneView &sc, const CameraPOD &cam, const RenderArgs &a, float *out, int *ids, unsigned long long *counters,
                              uint32_t blocks, hipStream_t stream)
{
    if constexpr (SD == 16) {
        if (a.node_cache == 3 && sc.nnodes <= CRT_WHOLE_TREE_NODES) {
            hipLaunchKernelGGL((render_kernel<CW, 16, SS, 16, CRT_WHOLE_TREE_NODES, RING>), dim3((blocks + 15u) / 16u), dim3(1024), 0, stream, sc, cam, a, out, ids, counters);
            return true;
        }
    }
    return false;
}

// One copy of the tree's top per CU instead of five.  Five 4-wave workgroups per CU hold five copies of the same ~270 nodes, and
// 80 of the CU's 160 KB of LDS are stacks.  A workgroup cannot have more than 16 waves, so a CU gets ONE 16-wave workgroup whose
// LDS copy holds the whole node array of a small tree (1328 nodes beside a 14-level stack, 1176 beside 16 levels) and ONE 4-wave
// workgroup with a small copy: two launches on two streams that take their work from the same block counters of the same frame,
// 16 + 4 = the same 20 waves per CU (both kernels at 96 registers: five waves per SIMD, see render_kernel_dyn).  On the headline frame no wave of the large
// workgroup ever fetches a node from device memory: 8.33 ms per step against 8.87.  Both launches are the
// same kernel with other template arguments, so the frame is the same bit for bit.  The small copy must stay below 32 KB per
// workgroup: above it the compiler sees an LDS-limited occupancy of 4 waves per SIMD, gives the kernel 128 registers, and the
// workgroup no longer fits beside the large one (16 ms).
#define CRT_NODE_CACHE_NODES_14 418 // 4 x 2 KB of stack (16-bit entries) + 418 x 56 B = 30.9 KB
#define CRT_SPLIT_A_NODES_14 1328 // 16 x 2 KB   of stack + 1328 x 56 B = 104.6 KB (16-bit stack entries)
#ifndef CRT_SPLIT_B_NODES_14
#define CRT_SPLIT_B_NODES_14 800  //  4 x 2 KB   of stack +  800 x 56 B =  51.8 KB (all of it dynamic LDS)
#endif
#define CRT_SPLIT_A_NODES_16 1176 // 16 x 4.5 KB of stack + 1176 x 56 B = 136.3 KB
#define CRT_SPLIT_B_NODES_16 76   //  4 x 4.5 KB of stack +   76 x 56 B =  22.2 KB
#define CRT_SPLIT_A_NODES_24 508  // 16 x 6.5 KB of stack +  508 x 56 B = 131.8 KB
#define CRT_SPLIT_B_NODES_24 16   //  4 x 6.5 KB of stack +   16 x 56 B =  26.9 KB
// Up to 104 KB of dynamic LDS per workgroup: allowed explicitly, once per kernel and device.
template <class Kernel>
static bool allow_dynamic_lds(Kernel kernel, size_t bytes, std::atomic<uint64_t> &done, int dev)
{
    if (dev >= 64 || (done.load(std::memory_order_relaxed) >> dev & 1u)) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    done.fetch_or(1ull << dev, std::memory_order_relaxed);
    return true;
}

// A two-launch frame whose launches went out but whose join could not be enqueued: the frame must not be launched again in another
// form (it is rendering), and the caller's stream does not wait for the helper's half -- launch_render returns this as its error.
static thread_local hipError_t g_join_error = hipSuccess;
static void join_split(hipStream_t stream, const SplitStreams *split)
{
    hipError_t e = hipEventRecord(split->join, split->helper);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, split->join, 0);
    if (e != hipSuccess) g_join_error = e;
}

// The kernels of a two-launch frame for a stack of SD levels and what LDS they ask for at launch (see launch_split).
template <bool CW, int SD, bool SS, bool RING>
struct SplitKernels {
    static constexpr int NA = SD == 14 ? CRT_SPLIT_A_NODES_14 : SD == 16 ? CRT_SPLIT_A_NODES_16 : CRT_SPLIT_A_NODES_24;
    static constexpr int NB = SD == 14 ? CRT_SPLIT_B_NODES_14 : SD == 16 ? CRT_SPLIT_B_NODES_16 : CRT_SPLIT_B_NODES_24;
    static constexpr size_t large_lds = dynamic_lds_bytes<SD, 16, NA, 1>(), small_lds = dynamic_lds_bytes<SD, 4, NB, 2>();
    // (a tree that fits the large copy as a whole gets the kernel without device-memory node steps; 24-level trees never do)
    static auto large(bool whole) { return whole ? &render_kernel_dyn<CW, SD, SS, 16, NA, RING, 1, SD != 24> : &render_kernel_dyn<CW, SD, SS, 16, NA, RING, 1, false>; }
    static auto small() { return &render_kernel_dyn<CW, SD, SS, 4, NB, RING, 2, false>; }
    // allowed their dynamic LDS on `dev`: at commit (prepare_split_kernels), so that no frame -- crt_render_noalloc's included --
    // has anything left to set up; launch_split asks again, which costs one load of a bit mask
    static bool allow(int dev)
    {
        static std::atomic<uint64_t> done[3];
        return allow_dynamic_lds(large(false), large_lds, done[0], dev) && allow_dynamic_lds(large(true), large_lds, done[1], dev) &&
               allow_dynamic_lds(small(), small_lds, done[2], dev);
    }
};

template <bool CW, int SD, bool SS, bool RING>
static bool launch_split(const SceneView &sc, const CameraPOD &cam, const RenderArgs &a, float *out, int *ids, unsigned long long *counters, uint32_t blocks,
                         hipStream_t stream, const SplitStreams *split)
{
    if constexpr (SD == 14 || SD == 16 || SD == 24) {
        if (!split || !split->helper) return false;
        constexpr int NA = SD == 14 ? CRT_SPLIT_A_NODES_14 : SD == 16 ? CRT_SPLIT_A_NODES_16 : CRT_SPLIT_A_NODES_24;
        constexpr int NB = SD == 14 ? CRT_SPLIT_B_NODES_14 : SD == 16 ? CRT_SPLIT_B_NODES_16 : CRT_SPLIT_B_NODES_24;
        constexpr int NBC = SD == 14 ? CRT_NODE_CACHE_NODES_14 : NB; // (the instrumented small workgroup: static LDS below 32 KB)
        const uint32_t cus = (blocks + 19u) / 20u; // blocks = persistent waves, 20 per CU
        if constexpr (!CW) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || !SplitKernels<CW, SD, SS, RING>::allow(dev)) return false;
        }
        // the small workgroups start behind whatever the caller's stream holds, the caller's stream goes on behind them
        if (hipEventRecord(split->fork, stream) != hipSuccess || hipStreamWaitEvent(split->helper, split->fork, 0) != hipSuccess) return false;
        RenderArgs b = a; // the small workgroups' waves come behind the large ones' in the frame's ring
        b.ring_wave_base = a.ring_wave_base + 16u * cus;
        if constexpr (CW) {
            // (the instrumented variant needs more registers than 96: it keeps the forms with all of their LDS declared)
            hipLaunchKernelGGL((render_kernel<CW, SD, SS, 16, NA, RING>), dim3(cus), dim3(1024), 0, stream, sc, cam, a, out, ids, counters);
            hipLaunchKernelGGL((render_kernel<CW, SD, SS, 4, NBC, RING>), dim3(cus), dim3(256), 0, split->helper, sc, cam, b, out, ids, counters);
        } else {
            typedef SplitKernels<CW, SD, SS, RING> K;
            hipLaunchKernelGGL(K::large(SD != 24 && sc.nnodes <= (uint32_t)NA), dim3(cus), dim3(1024), K::large_lds, stream, sc, cam, a, out, ids, counters);
            hipLaunchKernelGGL(K::small(), dim3(cus), dim3(256), K::small_lds, split->helper, sc, cam, b, out, ids, counters);
        }
        join_split(stream, split);
        return true;
    }
    return false;
}

// The lane kernel for a stack of SD levels: four waves per workgroup sharing an LDS copy of the top of the tree where the LDS
// has room for one beside five workgroups' stacks (SD = 14: 272 nodes, 16: 236, 24: 88), else single waves.
// (if constexpr: only the combinations that can be launched are instantiated)
#ifndef CRT_NODE_CACHE_NODES_24
#define CRT_NODE_CACHE_NODES_24 88  // 4 x 6.5 KB of stack + 88 x 56 B = 31.6 KB
#endif
template <bool CW, int SD, bool SS, bool RING>
static void launch_lanes(const SceneView &sc, const CameraPOD &cam, const RenderArgs &a, float *out, int *ids, unsigned long long *counters, uint32_t blocks,
                         hipStream_t stream, const SplitStreams *split)
{
    if (launch_whole_tree<CW, SD, SS, RING>(sc, cam, a, out, ids, counters, blocks, stream)) return;
    if (a.node_cache == 4 && launch_split<CW, SD, SS, RING>(sc, cam, a, out, ids, counters, blocks, stream, split)) return;
    if constexpr (SD == 14 || SD == 16 || SD == 24) {
        constexpr int NC = SD == 14 ? CRT_NODE_CACHE_NODES_14 : SD == 16 ? CRT_NODE_CACHE_NODES : CRT_NODE_CACHE_NODES_24;
        if (a.node_cache) {
            hipLaunchKernelGGL((render_kernel<CW, SD, SS, 4, NC, RING>), dim3((blocks + 3u) / 4u), dim3(256), 0, stream, sc, cam, a, out, ids, counters);
            return;
        }
    }
    hipLaunchKernelGGL((render_kernel<CW, SD, SS, 1, 0, RING>), dim3(blocks), dim3(64), 0, stream, sc, cam, a, out, ids, counters);
}

// Four-child nodes (SceneView.wide; crt_api.cpp widens the trees that are deep and large): the same forms -- 16 + 4 split, four-wave
// workgroups, single waves -- for the 16-level stack with 32-bit entries (an entry carries a four-bit child mask above the index,
// and a wide tree is at most half as deep as the two-child tree it was made from).
#define CRT_WIDE_A_NODES 584 // 16 x 4.5 KB of stack + 584 x 112 B = 135.9 KB
#define CRT_WIDE_B_NODES 38  //  4 x 4.5 KB of stack +  38 x 112 B =  22.2 KB
#define CRT_WIDE_C_NODES 118 //  4 x 4.5 KB of stack + 118 x 112 B =  31.2 KB (below 32 KB: see above)
template <bool CW, bool SS, bool RING>
struct WideKernels {
    static constexpr size_t large_lds = dynamic_lds_bytes<16, 16, CRT_WIDE_A_NODES, 1, true>(), small_lds = dynamic_lds_bytes<16, 4, CRT_WIDE_B_NODES, 2, true>();
    static auto large() { return &render_kernel_dyn<CW, 16, SS, 16, CRT_WIDE_A_NODES, RING, 1, false, true>; }
    static auto small() { return &render_kernel_dyn<CW, 16, SS, 4, CRT_WIDE_B_NODES, RING, 2, false, true>; }
    static bool allow(int dev)
    {
        static std::atomic<uint64_t> done[2];
        return allow_dynamic_lds(large(), large_lds, done[0], dev) && allow_dynamic_lds(small(), small_lds, done[1], dev);
    }
};
template <bool CW, bool SS, bool RING>
static void launch_wide(const SceneView &sc, const CameraPOD &cam, const RenderArgs &a, float *out, int *ids, unsigned long long *counters, uint32_t blocks,
                        hipStream_t stream, const SplitStreams *split)
{
    int dev = 0;
    if (a.node_cache == 4 && split && split->helper && hipGetDevice(&dev) == hipSuccess && (CW || WideKernels<CW, SS, RING>::allow(dev)) &&
        hipEventRecord(split->fork, stream) == hipSuccess && hipStreamWaitEvent(split->helper, split->fork, 0) == hipSuccess) {
        const uint32_t cus = (blocks + 19u) / 20u;
        RenderArgs b = a;
        b.ring_wave_base = a.ring_wave_base + 16u * cus;
        if constexpr (CW) {
            hipLaunchKernelGGL((render_kernel<CW, 16, SS, 16, CRT_WIDE_A_NODES, RING, true>), dim3(cus), dim3(1024), 0, stream, sc, cam, a, out, ids, counters);
            hipLaunchKernelGGL((render_kernel<CW, 16, SS, 4, CRT_WIDE_B_NODES, RING, true>), dim3(cus), dim3(256), 0, split->helper, sc, cam, b, out, ids, counters);
        } else {
            typedef WideKernels<CW, SS, RING> K;
            hipLaunchKernelGGL(K::large(), dim3(cus), dim3(1024), K::large_lds, stream, sc, cam, a, out, ids, counters);
            hipLaunchKernelGGL(K::small(), dim3(cus), dim3(256), K::small_lds, split->helper, sc, cam, b, out, ids, counters);
        }
        join_split(stream, split);
        return;
    }
    if (a.node_cache) {
        hipLaunchKernelGGL((render_kernel<CW, 16, SS, 4, CRT_WIDE_C_NODES, RING, true>), dim3((blocks + 3u) / 4u), dim3(256), 0, stream, sc, cam, a, out, ids, counters);
        return;
    }
    hipLaunchKernelGGL((render_kernel<CW, 16, SS, 1, 0, RING, true>), dim3(blocks), dim3(64), 0, stream, sc, cam, a, out, ids, counters);
}

// At commit, with the scene's device current: everything a two-launch frame would otherwise set up on its first use.
hipError_t prepare_split_kernels()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    bool ok = true;
#define CRT_ALLOW(SD, SS) ok = ok && SplitKernels<false, SD, SS, false>::allow(dev) && SplitKernels<false, SD, SS, true>::allow(dev)
    CRT_ALLOW(14, false); CRT_ALLOW(14, true); CRT_ALLOW(16, false); CRT_ALLOW(16, true); CRT_ALLOW(24, false); CRT_ALLOW(24, true);
#undef CRT_ALLOW
    ok = ok && WideKernels<false, false, false>::allow(dev) && WideKernels<false, false, true>::allow(dev) && WideKernels<false, true, false>::allow(dev) &&
         WideKernels<false, true, true>::allow(dev);
    return ok ? hipSuccess : hipErrorInvalidValue; // (not fatal for the caller: launch_split falls back to the one-launch form)
}

// The work queue of a frame: a.local_pixels, a.chunks and a.node_cache are final.  A block -- what a wave takes from a block counter
// in one trip -- is 2^group_shift pixel slots = block_units work units.  Larger blocks mean fewer trips, smaller ones a more even
// end of the frame; measured at one sample per unit (headline frame, tools/fetch_block_sweep.py, ms per frame at 128 / 256 / 512 /
// 1024 units): whole frame 9.56 / 9.46 / 9.42 / 9.49, 1/2 of it 4.86 / 4.79 / 4.79 / 4.86, 1/4 2.46 / 2.42 / 2.42 / 2.49, 1/8
// 1.267 / 1.254 / 1.268 / 1.34; the two-launch frame (16 + 4 split) wants at most 128: its 16-wave workgroups only leave a CU when
// all sixteen waves have run out of work (64 / 128 / 256 / 512: whole frame 8.60 / 8.37 / 8.39 / 8.61, 1/8 1.28 / 1.27 / 1.34 / 1.49).
// Whole pixels: a resident wave makes at least 8 trips per frame, blocks of 64..512 pixels (128 for the split).  Sample chunks: about
// 128 units whatever the form -- a block is also what a wave keeps in its ring until all of its units are in (crt_kernels.h), and
// at least 64 units so that one block covers a wave's request.
static uint32_t units_per_pixel(const RenderArgs &a) { return a.chunk_mode == CM_RING ? a.spp : a.chunk_mode == CM_SLAB ? a.chunks : 1u; }
static uint32_t chunk_group_shift(const RenderArgs &a)
{
    const uint32_t target = a.fetch_units ? (a.fetch_units < 64u ? 64u : a.fetch_units) : 128u, per_pixel = units_per_pixel(a);
    uint32_t shift = 0;
    while (shift < 6u && ((uint64_t)per_pixel << (shift + 1u)) <= target) shift++; // whole groups of pixels: at most 64 (a rank's pixel slots are a multiple of 64), at least 1
    while (shift < 6u && ((uint64_t)per_pixel << shift) < 64u) shift++;
    return shift;
}
static void plan_blocks(RenderArgs &a, uint32_t max_waves)
{
    a.chunk_spp = a.spp / a.chunks;
    a.per_pixel = units_per_pixel(a);
    a.per_pixel_magic = (uint32_t)(((1ull << 32) + a.per_pixel - 1u) / a.per_pixel);
    if (a.chunk_mode != CM_WHOLE) a.group_shift = chunk_group_shift(a);
    else {
        uint32_t shift = 0;
        if (a.fetch_units) { while ((2u << shift) <= a.fetch_units) shift++; }
        else {
            const uint64_t per_wave = (uint64_t)a.local_pixels / (max_waves ? max_waves : 1u);
            const uint32_t top = a.node_cache == 4 ? 7u : 9u;
            shift = 6;
            while (shift < top && (16ull << shift) <= per_wave) shift++;
        }
        a.group_shift = shift < 6u ? 6u : shift > 16u ? 16u : shift;
    }
    a.block_units = a.per_pixel << a.group_shift;
    a.nblocks = (uint32_t)((((uint64_t)a.local_pixels) + (1u << a.group_shift) - 1u) >> a.group_shift);
    if (a.blocks_per_stripe) a.blocks_per_stripe = (a.nblocks + CRT_WORK_STRIPES - 1u) / CRT_WORK_STRIPES; // contiguous bands: ceil(blocks / stripes)
    // the trip-counter guard of the kernel's main loop: a wave that did 64 times its share of the frame's path segments (the others
    // may be late) at 64 passes per segment must still finish; at least 2^24 passes, at most 2^32 - 1 (20 minutes at the very least)
    const uint64_t passes = ((uint64_t)a.local_pixels * a.per_pixel / (max_waves ? max_waves : 1u) + 1u) * 64u * (a.spp / a.per_pixel) * (a.max_depth ? a.max_depth : 1u) * 64u;
    a.guard_passes = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(passes, 1ull << 24), 0xFFFFFFFFull);
    if (a.ring_slots == 0) a.ring_slots = CRT_RING_SLOTS_DEFAULT;
    a.ring_slots = a.ring_slots < 2u ? 2u : a.ring_slots > 16u ? 16u : a.ring_slots;
    if (a.ring_batch == 0) a.ring_batch = CRT_RING_BATCH_DEFAULT;
    a.ring_batch = a.ring_batch > a.ring_slots / 2u ? a.ring_slots / 2u : a.ring_batch; // (half of the ring stays for blocks in flight)
}

// (a workgroup form may round the persistent waves up to a multiple of 16)
size_t render_ring_bytes(const RenderArgs &ra, uint32_t max_waves, uint32_t slots)
{
    if (ra.chunk_mode != CM_RING) return 0;
    return (size_t)(max_waves + 16u) * slots * ((size_t)ra.spp << chunk_group_shift(ra)) * 3u * sizeof(float);
}
uint32_t render_ring_slots_that_fit(const RenderArgs &ra, uint32_t max_waves, size_t budget)
{
    if (ra.chunk_mode != CM_RING) return CRT_RING_SLOTS_DEFAULT;
    uint32_t slots = ra.ring_slots ? ra.ring_slots : CRT_RING_SLOTS_DEFAULT;
    slots = slots < 2u ? 2u : slots > 16u ? 16u : slots;
    while (slots >= 2u && render_ring_bytes(ra, max_waves, slots) > budget) slots--;
    return slots >= 2u ? slots : 0u;
}

hipError_t launch_render(const SceneView &sc, const CameraPOD &cam, const RenderArgs &ra, float *out, int *ids, unsigned long long *counters,
                         bool count_work, uint32_t max_waves, hipStream_t stream, const SplitStreams *split)
{
    const uint32_t ntiles = ra.tiles_x * ra.tiles_y;
    const uint32_t local_tiles = (ntiles + ra.tile_ranks - 1 - ra.tile_rank) / ra.tile_ranks; // tiles k = b*ranks + rank < ntiles
    if (local_tiles == 0) return hipSuccess;
    g_join_error = hipSuccess;
    RenderArgs a = ra;
    a.local_pixels = local_tiles * 64u;
    if (a.chunks <= 1) { a.chunks = 1; a.chunk_mode = CM_WHOLE; }
    else if (a.chunk_mode == CM_WHOLE) a.chunk_mode = CM_RING;
    const uint64_t units = (uint64_t)local_tiles * units_per_pixel(a); // wave-sized work units
    const uint32_t blocks = units < max_waves ? (uint32_t)units : max_waves;
    // node_cache = 1, chosen by the frame: one that does not fill the GPU spreads better as single waves (0); one that fills it many
    // times over takes the 16 + 4 split where the tree allows it (4: the large workgroups' LDS copies and their 16 waves that
    // must all finish before the slot is free only pay on long frames); else four-wave workgroups (2)
    if (a.node_cache == 1) a.node_cache = units < max_waves ? 0u : (units >= 8ull * max_waves && split && split->helper && ra.bvh_depth <= 24 && blocks == max_waves) ? 4u : 2u;
    if (a.node_cache == 4 && !(split && split->helper && ra.bvh_depth <= 24)) a.node_cache = 2; // (asked for by name: small frames too, the tests do)
    plan_blocks(a, max_waves);
    a.tiles_x_magic = a.tiles_x > 1 ? (uint32_t)((1ull << 32) / a.tiles_x) : 0xFFFFFFFFu; // tiles_x == 1: tile / 1 = tile - (0 or 1), fixed up
    // persistent waves (blocks, above): enough to fill the device; each lane keeps drawing pixels until the counter runs out
#ifdef CRT_WITH_POOL_KERNEL
    if (ra.kernel == RK_POOL && a.chunk_mode == CM_WHOLE) { // (the pool kernel has no sample chunks)
        // Experimental (CRT_KERNEL=1).  One wave per workgroup, 96 contexts, no LDS node cache: the variant that measured
        // best (8 waves sharing an LDS copy of the whole node array ran slower, 1.75 waves per SIMD
        // cannot keep the VALU busy).
#define CRT_LAUNCH_POOL(CW, PP, SD, WW, NC) \
    hipLaunchKernelGGL((render_pool_kernel<CW, PP, SD, WW, NC>), dim3(blocks), dim3(64 * WW), 0, stream, sc, cam, a, out, ids, counters)
        const bool shallow = ra.bvh_depth <= 16;
        if (count_work) { if (shallow) CRT_LAUNCH_POOL(true, 96, 16, 1, 0); else CRT_LAUNCH_POOL(true, 96, 32, 1, 0); }
        else { if (shallow) CRT_LAUNCH_POOL(false, 96, 16, 1, 0); else CRT_LAUNCH_POOL(false, 96, 32, 1, 0); }
#undef CRT_LAUNCH_POOL
        return hipGetLastError();
    }
#else
    if (ra.kernel == RK_POOL && a.chunk_mode == CM_WHOLE) return hipErrorNotSupported; // (crt_api.cpp only asks for it when pool_kernel_built())
#endif
    // the LDS stack is sized by the tree: 14 levels (4 KB per wave), 16 (4.5 KB), 24 (6.5 KB: still the 20 waves per CU the registers
    // allow) or 32 (8.5 KB: 18 waves per CU)
    const bool ss = sc.sphere_slots != 0;
    // (the whole-tree variant exists for the 16-level stack only)
    // (14 levels: the kernels with 16-bit stack entries, for trees whose references fit a short -- LdsStack16)
    const bool short_refs = sc.nnodes < 32767u && sc.ntris <= 4096u;
    const int levels = (ra.bvh_depth <= 14 && short_refs && a.node_cache != 3) ? 14 : ra.bvh_depth <= 16 ? 16 : ra.bvh_depth <= 24 ? 24 : CRT_STACK_DEPTH;
#define CRT_LAUNCH_RING(CW, SD, SS) do { if (a.chunk_mode == CM_RING) launch_lanes<CW, SD, SS, true>(sc, cam, a, out, ids, counters, blocks, stream, split); \
                                         else launch_lanes<CW, SD, SS, false>(sc, cam, a, out, ids, counters, blocks, stream, split); } while (0)
#define CRT_LAUNCH_SS(CW, SD) do { if (ss) CRT_LAUNCH_RING(CW, SD, true); else CRT_LAUNCH_RING(CW, SD, false); } while (0)
#define CRT_LAUNCH_SD(CW) do { if (levels == 14) CRT_LAUNCH_SS(CW, 14); else if (levels == 16) CRT_LAUNCH_SS(CW, 16); else if (levels == 24) CRT_LAUNCH_SS(CW, 24); \
                               else CRT_LAUNCH_SS(CW, CRT_STACK_DEPTH); } while (0)
    if (sc.wide) {
        if (ra.bvh_depth > 16) return hipErrorInvalidValue; // (crt_api.cpp only widens trees that end up within the 16-level stack)
        if (a.node_cache == 3) a.node_cache = 2;
#define CRT_LAUNCH_WIDE(CW, SS) do { if (a.chunk_mode == CM_RING) launch_wide<CW, SS, true>(sc, cam, a, out, ids, counters, blocks, stream, split); \
                                     else launch_wide<CW, SS, false>(sc, cam, a, out, ids, counters, blocks, stream, split); } while (0)
        if (count_work) { if (ss) CRT_LAUNCH_WIDE(true, true); else CRT_LAUNCH_WIDE(true, false); }
        else { if (ss) CRT_LAUNCH_WIDE(false, true); else CRT_LAUNCH_WIDE(false, false); }
#undef CRT_LAUNCH_WIDE
        return g_join_error != hipSuccess ? g_join_error : hipGetLastError();
    }
    if (count_work) CRT_LAUNCH_SD(true); else CRT_LAUNCH_SD(false);
#undef CRT_LAUNCH_SD
#undef CRT_LAUNCH_SS
#undef CRT_LAUNCH_RING
    return g_join_error != hipSuccess ? g_join_error : hipGetLastError();
}

bool pool_kernel_built()
{
#ifdef CRT_WITH_POOL_KERNEL
    return true;
#else
    return false;
#endif
}

hipError_t launch_untile(const float *compact_all, uint32_t width, uint32_t height, uint32_t tile_ranks, float *out, hipStream_t stream)
{
    const uint32_t tiles_x = (width + 7u) / 8u, tiles_y = (height + 7u) / 8u, ntiles = tiles_x * tiles_y;
    const uint32_t tiles_per_rank = (ntiles + tile_ranks - 1) / tile_ranks;
    hipLaunchKernelGGL(untile_kernel, dim3(ntiles), dim3(64), 0, stream, compact_all, width, height, tiles_x, ntiles, tile_ranks, tiles_per_rank, out);
    return hipGetLastError();
}

// utility.h:133-137: (unsigned char) min(max(x * 255.0f, 0.0f), 255.0f); four values per lane, one dword store
__global__ __launch_bounds__(256) void quantize_kernel(const float *__restrict__ rgb, size_t n, uint8_t *__restrict__ out)
{
    const size_t k4 = ((size_t)blockIdx.x * 256u + threadIdx.x) * 4u;
    if (k4 >= n) return;
    uint32_t b[4] = { 0, 0, 0, 0 };
    const uint32_t m = n - k4 < 4u ? (uint32_t)(n - k4) : 4u;
    for (uint32_t k = 0; k < m; k++) {
        float v = rgb[k4 + k] * 255.0f;
        v = __builtin_fmaxf(v, 0.0f); // std::max(x, 0): a NaN x gives 0
        v = __builtin_fminf(v, 255.0f);
        b[k] = (uint32_t)v;
    }
    if (m == 4u && ((uintptr_t)(out + k4) & 3u) == 0) *reinterpret_cast<uint32_t *>(out + k4) = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    else for (uint32_t k = 0; k < m; k++) out[k4 + k] = (uint8_t)b[k];
}

hipError_t launch_quantize(const float *rgb, size_t n, uint8_t *out, hipStream_t stream)
{
    const uint32_t blocks = (uint32_t)((n + 1023) / 1024);
    hipLaunchKernelGGL(quantize_kernel, dim3(blocks), dim3(256), 0, stream, rgb, n, out);
    return hipGetLastError();
}

hipError_t launch_finalize(const float *accum, size_t n, uint32_t total_spp, float *out, hipStream_t stream)
{
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(256), 0, stream, accum, n, (float)total_spp, out);
    return hipGetLastError();
}

} // namespace crt
