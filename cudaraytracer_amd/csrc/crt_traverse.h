// crt_traverse.h -- closest-hit search over the 64-byte two-child BVH and the reference-order tie fold,
// plus the per-sample path loop.  Per-lane code, templated on the stack storage (LDS columns on gfx950, a
// plain array in the host single-stepper) and on the work counters.
//
// Why a different BVH and a different visit order still give the reference's answer
// ---------------------------------------------------------------------------------
// HittableList::hit (hittable_list.h:32-60) walks its LBVH depth-first, children[0] first, never culls by
// distance (aabb.h:72-92 overwrites the interval it is given, F8) and lets a leaf replace the current hit
// when  t <= closest + 1e-7  (triangle.h:75) or  t <= closest  (sphere.h:36, interval.h:19-21).  Its DFS leaf
// order is the stable Morton order (bvh.h:120-157), called `rank` here.  So the reference result is a fold, in
// rank order, over the primitives the ray actually hits, and each primitive's own t does not depend on
// closest_so_far (crt_core.h: triangle_t, sphere_t).  Let t_min be the smallest candidate t and m the first
// primitive in rank order that attains it: whatever happened before m, m is accepted when visited, and after m
// only candidates within a few 1e-7 steps of t_min can be accepted.  Therefore:
//   * if no other candidate lies within W = 1e-6 of t_min, the reference returns m            (fast path);
//   * otherwise fold, in rank order, the candidates with t <= t_min + W with the reference's own
//     comparisons                                                                             (tie path).
// The fast path is an ordinary nearest-hit traversal (near child first, distance culling) that also raises a
// flag whenever two candidates come within W of each other; the tie path re-walks the tree once per
// candidate in the window, each time extracting the next rank, so it needs no storage.  It runs for a few
// rays per million.  Box tests only prune: boxes contain the reference's own 1e-4-padded leaf boxes
// (triangle.h:28-41) plus a margin that covers the rounding of the slab arithmetic below (crt_bvh.cpp).
//
// One case is EXCLUDED from this argument, because there the reference's box test is not "purely geometric":
// a direction component of exactly +-0 together with an origin that lies, in that component and bit for bit, on
// a plane of one of the reference's node boxes.  aabb.h:73-78 then forms 0 * inf = NaN, and glm's min / max
// (first operand returned when either is NaN) turn it into a miss of that box -- always on a max plane, on a
// min plane for the x axis only (tests/test_oracle.py::test_slab_test_nan_case_depends_on_the_axis records what
// the restatement returns) -- although the ray runs inside the box's closed slab, so the reference skips a
// subtree whose primitives this walk would test (here the reciprocal is clamped to +-1e30 and 0 * 1e30 = 0).
// Generated rays do not reach it: a camera direction is pixel_center - origin (camera.h:96) and a scattered one
// normal + random_unit_vector (material.h:25), whose components vanish only by exact cancellation, and the
// origin -- a hit point o + t d -- would in addition have to equal a coordinate of a 1e-4-padded triangle box.
#pragma once

#include "crt_core.h"

namespace crt {

struct alignas(16) q4 { float x, y, z, w; };

CRT_HD int f2i(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_int(f);
#else
    union { float f; int i; } u; u.f = f; return u.i;
#endif
}
CRT_HD float i2f(int i)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(i);
#else
    union { float f; int i; } u; u.i = i; return u.f;
#endif
}

// Node (4 x q4 = 64 B): [c0min.xyz, ref0] [c0max.xyz, ref1] [c1min.xyz, -] [c1max.xyz, -]
// ref >= 0: index of a node; ref < 0: leaf, ~ref = (first_slot << 3) | (count - 1), count in 1..8.
// Triangle slot (3 x q4 = 48 B): [a.xyz, prim id] [b.xyz, rank] [c.xyz, material]
//
// Four-child node (SceneView.wide; 8 x q4 = 128 B, 7 of them read): [min.x of children 0..3] [min.y] [min.z] [max.x] [max.y] [max.z]
// [ref 0..3] [-].  Half as many dependent fetches per walk for the same box arithmetic: the format for trees whose nodes do not
// fit the caches (the 1 M-triangle grid), where a walk waits for memory between node steps.  A reference to such a node carries, above
// its index, the mask of the children to test (CRT_WIDE_SHIFT): all that exist in the references the builder writes; and when
// three or four children of a node are hit the walk stores ONE stack entry -- the node itself with the mask of the children still
// to visit -- and tests those again when it comes back (against a limit that has shrunk since), so the stack holds one entry per
// level of the wide tree like the two-child walk's, whatever the fan-out.
#define CRT_WIDE_SHIFT 26
#define CRT_WIDE_INDEX 0x03FFFFFF
struct SceneView {
    const q4 *nodes;
    const q4 *tris;
    const Sphere *spheres;
    const Shade *shades; // per material id
    const q4 *normals;   // per triangle slot: unit normal of the triangle as triangle.h:81 computes it, primitive id in w
    int root_ref;        // BVH_NONE when there are no triangles
    uint32_t nnodes;     // 64-byte nodes (breadth-first order: crt_bvh.cpp)
    uint32_t ntris;      // triangle slots
    uint32_t nspheres;   // flat list
    uint32_t sphere_slots; // != 0: some slots of `tris` hold spheres (picks the kernel variant that tests for them); the bits of a float: the largest radius among them (at least FLT_MIN)
    uint32_t wide;         // != 0: `nodes` holds four-child nodes of 128 bytes (trav_inner4_from) instead of two-child nodes of 64
    f3 root_min, root_max;
};

#define CRT_BVH_NONE 0x7FFFFFFF
// bit 30 of a slot's rank word: the slot holds a sphere ([centre, id][radius argument, 1 / radius or 0, sphere_sure_term, rank][radius, -, -, material]); the same
// bit in a hit code says "sphere slot" (codes: triangle slot < ntris <= flat sphere < 2^30 <= sphere slot)
#define CRT_SPHERE_SLOT 0x40000000
#define CRT_TIE_WINDOW 1e-6f
#define CRT_STACK_DEPTH 32 // crt_bvh.cpp keeps the tree depth below this

// A counters type also says, at compile time, whether leaves may hold sphere slots: triangle-only scenes (every BASELINE
// configuration) run kernels without that test (it costs 7 % on the headline frame when compiled in).
struct NoCounters {
    static constexpr bool sphere_slots = true;
    CRT_HD void node() {}
    CRT_HD void tri() {}
    CRT_HD void sphere() {}
    CRT_HD void ray() {}
    CRT_HD void refold() {}
    CRT_HD void shade() {}
};

// MODE_CLOSEST: the nearest-hit walk of the common path (sphere hits decided by sphere_hit_implies_box or taken provisionally, see
// crt_core.h: sphere_t).  MODE_NEXT_RANK: the careful path's extraction walk, with the reference's literal box test for every
// sphere hit.
enum { MODE_CLOSEST = 0, MODE_NEXT_RANK = 1 };

// State of one search.  code: triangle slot, or ntris + sphere index; -1 = nothing.
struct Search {
    float t;       // MODE_CLOSEST: smallest t so far.  MODE_NEXT_RANK: t of the candidate found
    int code;
    int mat;       // material id of `code` (carried so that the shading record is fetched without waiting for the primitive)
    int rank;      // MODE_NEXT_RANK: rank of the candidate found (smallest rank > after_rank)
    int near_tie;  // MODE_CLOSEST: 1 when two candidates within CRT_TIE_WINDOW of each other were seen, or a sphere hit was taken provisionally (an int: a lane
                   // mask would be merged with scalar instructions at every join of the walk loop).  MODE_NEXT_RANK: 0, or -1 to
                   // search for the largest rank instead (`rank` then holds ~rank, and after_rank must be INT_MIN)
};

template <int MODE>
CRT_HD void offer(Search &s, float t, int code, int mat, int rank, int after_rank, float t_hi)
{
    if (MODE == MODE_CLOSEST) {
        s.near_tie = (s.code >= 0 && __builtin_fabsf(t - s.t) <= CRT_TIE_WINDOW) ? 1 : s.near_tie;
        if (t < s.t) { s.t = t; s.code = code; s.mat = mat; }
    } else {
        // (s.near_tie, unused in this mode, is the order mask: 0 = smallest rank, -1 = the comparison reversed, i.e. the LARGEST rank)
        const int key = rank ^ s.near_tie;
        if (t <= t_hi && key > after_rank && key < s.rank) { s.t = t; s.code = code; s.mat = mat; s.rank = key; }
    }
}

// Traversal registers of one lane.  The walk is cut into steps (begin / inner node / leaf) so that the gfx950
// kernel can interleave lanes that are at different stages (crt_kernels.hip); walk() below strings them together.
struct Trav {
    f3 idir, ood; // slab arithmetic: t' = b * idir - ood, origin re-based to the root-box entry
    float shift;  // t = t' + shift
    float hi;     // upper pruning bound in re-based time
    float slack;  // sphere-slot scenes only (prune_limit): what a sphere's computed root may precede its box by, beyond the relative part
    int cur;      // current ref (node >= 0, leaf < 0) or CRT_BVH_NONE when the walk is over
};

// min(a, b) for operands that are never NaN.  In IEEE mode the compiler has to quiet a possible signalling NaN before every
// v_min / v_max whose operand it cannot trace to an arithmetic result (`v_max_f32 x, x, x`): one instruction per node step
// for the walk's far limit, which lives in a register across the whole state machine.  (Pruning arithmetic only.)
CRT_HD float min_no_nan(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __builtin_fminf(a, b);
#endif
}

// x limited to [-1e30, 1e30] (x is never NaN): one v_med3_f32
CRT_HD float clamp_1e30(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, -1e30f, 1e30f);
#else
    return __builtin_fminf(__builtin_fmaxf(x, -1e30f), 1e30f);
#endif
}

CRT_HD float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// element k of a read-only device array, k the same in every lane: on the device through the constant address space, so
// that it is a scalar load (one request through the scalar cache instead of a vector load that keeps the CU's address
// unit busy for 16 cycles per 16 bytes)
template <class T>
CRT_HD T load_uniform(const T *p, uint32_t k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const T *constant_ptr;
    return *((constant_ptr)(uintptr_t)p + k);
#else
    return p[k];
#endif
}

// element k of a device array through the global address space (a pointer rebuilt from raw kernel-argument words is generic
// to the compiler, and generic means flat_load)
template <class T>
CRT_HD T load_global(const T *p, size_t k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(1))) const T *global_ptr;
    return *((global_ptr)(uintptr_t)p + k);
#else
    return p[k];
#endif
}

// The end-of-walk value of a stack type: CRT_BVH_NONE, or a smaller one where the stack's entries are narrower than an int
// (crt_kernels.hip: the 16-bit stack of small trees; every reference of such a tree lies below it).
template <class Stack> struct stack_none { static constexpr int value = CRT_BVH_NONE; };
template <int NONE = CRT_BVH_NONE>
CRT_HD bool is_inner(int ref) { return (unsigned)ref < (unsigned)NONE; }

// How far the walk may prune.  A triangle's t lies inside the time interval of its boxes (padded by 1e-4), so
// `best` + the tie window is the limit.  A sphere's computed root carries the cancellation of its discriminant and can lie BEFORE
// the entry time of the sphere's own box: (4u |oc| + sqrt(u (19 |oc|^2 + 6 r^2))) / |d| <= 1.1e-3 (t_c + r / |d|) with t_c the time
// of the closest approach to the centre (crt_core.h: the bounds of sphere_hit_implies_box).  Kernels for scenes with spheres in the
// tree therefore prune at best * (1 + CRT_SPHERE_ROOT_REL) + slack, slack = CRT_SPHERE_ROOT_ABS * r_max / |d| per ray.  (That covers
// roots of rays that do intersect the sphere; for a hit the arithmetic invents just outside the silhouette it is a mitigation, see
// DESIGN.md section 2 "What remains unmatched".)
#define CRT_SPHERE_ROOT_REL 2.5e-3f
#define CRT_SPHERE_ROOT_ABS 1.0e-2f
template <bool SS>
CRT_HD float prune_limit(const SceneView &sc, const Trav &tv, float best)
{
    if (SS && sc.sphere_slots) return fma1(best, 1.0f + CRT_SPHERE_ROOT_REL, tv.slack);
    return best;
}

// Spheres: a short flat list (ground + light in the reference scenes, scene.h:164-165), tested before the BVH.
// MODE_CLOSEST: s.t must start at tmax (FLT_MAX).  MODE_NEXT_RANK: only candidates with t <= t_hi and
// rank > after_rank are considered, s.rank must start at INT_MAX.
template <int MODE, class Counters>
CRT_HD void trav_spheres(const SceneView &sc, f3 o, f3 d, float tmin, Search &s, int after_rank, float t_hi, Counters &cnt)
{
    for (uint32_t k = 0; k < sc.nspheres; k++) {
        const Sphere sp = load_uniform(sc.spheres, k);
        float t;
        cnt.sphere();
        bool doubt = false;
        if (sphere_t<MODE != MODE_CLOSEST>(mk3(sp.cx, sp.cy, sp.cz), sp.radius, sp.radius_arg, sp.sure, o, d, tmin, t, doubt)) {
            offer<MODE>(s, t, (int)(sc.ntris + k), sp.mat, sp.rank, after_rank, t_hi);
            if (MODE == MODE_CLOSEST) s.near_tie = doubt ? 1 : s.near_tie;
        }
    }
}

// Root box of the triangle BVH against the current best t: sets up the slab registers.  Leaves
// tv.cur = CRT_BVH_NONE if no node needs visiting.
template <int MODE, int NONE = CRT_BVH_NONE, bool SS = true>
CRT_HD void trav_root(const SceneView &sc, f3 o, f3 d, const Search &s, float t_hi, Trav &tv)
{
    tv.cur = NONE;
    if (sc.root_ref == CRT_BVH_NONE) return;
    if (SS && sc.sphere_slots) tv.slack = CRT_SPHERE_ROOT_ABS * i2f((int)sc.sphere_slots) / __builtin_sqrtf(dot(d, d));
    // slab arithmetic for pruning: idir finite, origin re-based to the root box entry so that rounding stays
    // small against the box margin wherever the ray starts.
    // (pruning only, so the 1-ulp hardware reciprocal is enough; it never touches a result)
    const f3 idir = mk3(clamp_1e30(fast_rcp(d.x)), clamp_1e30(fast_rcp(d.y)), clamp_1e30(fast_rcp(d.z))); // 1 / +-0 = +-inf -> +-1e30
    const float ax = (sc.root_min.x - o.x) * idir.x, bx = (sc.root_max.x - o.x) * idir.x;
    const float ay = (sc.root_min.y - o.y) * idir.y, by = (sc.root_max.y - o.y) * idir.y;
    const float az = (sc.root_min.z - o.z) * idir.z, bz = (sc.root_max.z - o.z) * idir.z;
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    const float limit = prune_limit<SS>(sc, tv, (MODE == MODE_CLOSEST) ? s.t : t_hi);
    const float bound = limit + CRT_TIE_WINDOW;
    if (!(tn <= tf && tf >= 0.0f && tn <= bound)) return;
    tv.shift = __builtin_fmaxf(tn, 0.0f); // (the same trick here, min_no_nan's counterpart, measured 2.8 % SLOWER: the asm pins the schedule of the shade phase)
    const f3 ob = axpy(tv.shift, d, o);
    tv.idir = idir;
    tv.ood = mk3(ob.x * idir.x, ob.y * idir.y, ob.z * idir.z);
    tv.hi = limit - tv.shift + CRT_TIE_WINDOW;
    tv.cur = sc.root_ref;
}

template <int MODE, class Stack, class Counters>
CRT_HD void trav_begin(const SceneView &sc, f3 o, f3 d, float tmin, Search &s, int after_rank, float t_hi, Trav &tv, Stack &stack, Counters &cnt)
{
    trav_spheres<MODE>(sc, o, d, tmin, s, after_rank, t_hi, cnt);
    trav_root<MODE, stack_none<Stack>::value, Counters::sphere_slots>(sc, o, d, s, t_hi, tv);
    stack.reset();
}

// Where node records come from: device memory, or (crt_kernels.hip) an LDS copy of the top of the tree.
struct GlobalNodes {
    const q4 *nodes;
    CRT_HD void fetch(int cur, q4 &n0, q4 &n1, q4 &n2, q4 &n3) const
    {
        const q4 *n = nodes + 4u * (uint32_t)cur;
        n0 = n[0]; n1 = n[1]; n2 = n[2]; n3 = n[3];
    }
};

// The end of a node step, branch-free: entry / exit times of the two child boxes -> near child next and far child pushed
// (both hit), the one child that was hit, or a pop (`below` = what a pop returns, read ahead by the caller).  The child that
// is not taken is stored to the stack top unconditionally: when nothing is pushed the store lands on the spare level and its
// value does not matter, so "next" and "stored" are two selects on ONE mask.  On the device the three comparisons are
// combined as 64-bit scalar masks (s_and / s_or / s_orn2): written as bool expressions the compiler evaluates `!(a <= b)`
// with a second compare each (NaN-safe, but one more instruction of the 4-cycle class per negation).
template <class Stack>
CRT_HD void trav_descend(float tn0, float tf0, float tn1, float tf1, int r0, int r1, int below, Trav &tv, Stack &stack)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(tn0 <= tf0), m1 = __builtin_amdgcn_ballot_w64(tn1 <= tf1);
    const unsigned long long msw = __builtin_amdgcn_ballot_w64(tn1 < tn0);
    const bool both = __builtin_amdgcn_inverse_ballot_w64(m0 & m1), none = __builtin_amdgcn_inverse_ballot_w64(~(m0 | m1));
    const bool take1 = __builtin_amdgcn_inverse_ballot_w64(~m0 | (m1 & msw));
#else
    const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1, swap = tn1 < tn0;
    const bool both = h0 && h1, none = !h0 && !h1;
    const bool take1 = !h0 || (h1 && swap); // = both ? swap : !h0
#endif
    const int next = take1 ? r1 : r0, other = take1 ? r0 : r1;
    stack.push_if(other, both);
    stack.pop_if(none);
    tv.cur = none ? below : next;
}

// One two-child node: both child boxes, near child next, far child pushed.  Requires is_inner(tv.cur).
template <class Nodes, class Stack, class Counters>
CRT_HD void trav_inner_from(const Nodes &src, Trav &tv, Stack &stack, Counters &cnt)
{
    q4 n0, n1, n2, n3;
    src.fetch(tv.cur, n0, n1, n2, n3);
    const int below = stack.peek(); // what a pop would return: read now, beside the node, used only if no child is hit
    cnt.node();
    const f3 idir = tv.idir, ood = tv.ood;
    const float lo = -tv.shift, hi = tv.hi; // t >= 0 in re-based time
    const float a0x = fma1(n0.x, idir.x, -ood.x), b0x = fma1(n1.x, idir.x, -ood.x);
    const float a0y = fma1(n0.y, idir.y, -ood.y), b0y = fma1(n1.y, idir.y, -ood.y);
    const float a0z = fma1(n0.z, idir.z, -ood.z), b0z = fma1(n1.z, idir.z, -ood.z);
    const float a1x = fma1(n2.x, idir.x, -ood.x), b1x = fma1(n3.x, idir.x, -ood.x);
    const float a1y = fma1(n2.y, idir.y, -ood.y), b1y = fma1(n3.y, idir.y, -ood.y);
    const float a1z = fma1(n2.z, idir.z, -ood.z), b1z = fma1(n3.z, idir.z, -ood.z);
    const float tn0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a0x, b0x), __builtin_fminf(a0y, b0y)),
                                      __builtin_fmaxf(__builtin_fminf(a0z, b0z), lo));
    const float tf0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a0x, b0x), __builtin_fmaxf(a0y, b0y)),
                                      min_no_nan(__builtin_fmaxf(a0z, b0z), hi));
    const float tn1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a1x, b1x), __builtin_fminf(a1y, b1y)),
                                      __builtin_fmaxf(__builtin_fminf(a1z, b1z), lo));
    const float tf1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a1x, b1x), __builtin_fmaxf(a1y, b1y)),
                                      min_no_nan(__builtin_fmaxf(a1z, b1z), hi));
    trav_descend(tn0, tf0, tn1, tf1, f2i(n0.w), f2i(n1.w), below, tv, stack);
}

template <class Stack, class Counters>
CRT_HD void trav_inner(const SceneView &sc, Trav &tv, Stack &stack, Counters &cnt)
{
    GlobalNodes src;
    src.nodes = sc.nodes;
    trav_inner_from(src, tv, stack, cnt);
}

struct GlobalNodes4 {
    const q4 *nodes;
    CRT_HD void fetch(uint32_t idx, q4 &mnx, q4 &mny, q4 &mnz, q4 &mxx, q4 &mxy, q4 &mxz, q4 &ref) const
    {
        const q4 *n = nodes + 8u * idx;
        mnx = n[0]; mny = n[1]; mnz = n[2]; mxx = n[3]; mxy = n[4]; mxz = n[5]; ref = n[6];
    }
};

// entry / exit time of one child box (pairwise form, as trav_inner_from)
CRT_HD void slab_times(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, f3 idir, f3 ood, float lo, float hi, float &tn, float &tf)
{
    const float ax = fma1(mnx, idir.x, -ood.x), bx = fma1(mxx, idir.x, -ood.x);
    const float ay = fma1(mny, idir.y, -ood.y), by = fma1(mxy, idir.y, -ood.y);
    const float az = fma1(mnz, idir.z, -ood.z), bz = fma1(mxz, idir.z, -ood.z);
    tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fmaxf(__builtin_fminf(az, bz), lo));
    tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), min_no_nan(__builtin_fmaxf(az, bz), hi));
}

// One four-child node: the children named by the reference's mask are tested, the nearest one that is hit comes next; one other
// hit child is pushed as it is, two or three as ONE entry (this node, their mask).  Requires is_inner(tv.cur).  Branch-free.
template <class Nodes, class Stack, class Counters>
CRT_HD void trav_inner4_from(const Nodes &src, Trav &tv, Stack &stack, Counters &cnt)
{
    const uint32_t idx = (uint32_t)tv.cur & CRT_WIDE_INDEX, mask = (uint32_t)tv.cur >> CRT_WIDE_SHIFT;
    q4 mnx, mny, mnz, mxx, mxy, mxz, rf;
    src.fetch(idx, mnx, mny, mnz, mxx, mxy, mxz, rf);
    const int below = stack.peek();
    cnt.node();
    const f3 idir = tv.idir, ood = tv.ood;
    const float lo = -tv.shift, hi = tv.hi;
    float tn0, tf0, tn1, tf1, tn2, tf2, tn3, tf3;
    slab_times(mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x, idir, ood, lo, hi, tn0, tf0);
    slab_times(mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y, idir, ood, lo, hi, tn1, tf1);
    slab_times(mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z, idir, ood, lo, hi, tn2, tf2);
    slab_times(mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w, idir, ood, lo, hi, tn3, tf3);
    const bool h0 = tn0 <= tf0 && (mask & 1u), h1 = tn1 <= tf1 && (mask & 2u), h2 = tn2 <= tf2 && (mask & 4u), h3 = tn3 <= tf3 && (mask & 8u);
    const uint32_t hits = (h0 ? 1u : 0u) | (h1 ? 2u : 0u) | (h2 ? 4u : 0u) | (h3 ? 8u : 0u);
    const int r0 = f2i(rf.x), r1 = f2i(rf.y), r2 = f2i(rf.z), r3 = f2i(rf.w);
    // the nearest hit child (the lowest index among equals)
    const float k0 = h0 ? tn0 : FLT_MAX, k1 = h1 ? tn1 : FLT_MAX, k2 = h2 ? tn2 : FLT_MAX, k3 = h3 ? tn3 : FLT_MAX;
    const bool b01 = k1 < k0, b23 = k3 < k2;
    const float k01 = b01 ? k1 : k0, k23 = b23 ? k3 : k2;
    const int r01 = b01 ? r1 : r0, r23 = b23 ? r3 : r2;
    const uint32_t i01 = b01 ? 2u : 1u, i23 = b23 ? 8u : 4u; // as bits
    const bool b = k23 < k01 || !(hits & 3u); // (children 0 and 1 both missed: FLT_MAX against FLT_MAX must not pick them)
    const int next = b ? r23 : r01;
    const uint32_t pending = hits & ~(b ? i23 : i01);
    const bool one = (pending & (pending - 1u)) == 0u; // none or one
    const int other = (pending & 1u) ? r0 : (pending & 2u) ? r1 : (pending & 4u) ? r2 : r3;
    const int entry = one ? other : (int)((pending << CRT_WIDE_SHIFT) | idx);
    stack.push_if(entry, pending != 0u);
    stack.pop_if(hits == 0u);
    tv.cur = hits == 0u ? below : next;
}

template <class Stack, class Counters>
CRT_HD void trav_inner4(const SceneView &sc, Trav &tv, Stack &stack, Counters &cnt)
{
    GlobalNodes4 src;
    src.nodes = sc.nodes;
    trav_inner4_from(src, tv, stack, cnt);
}

// One leaf: its 1..8 triangle slots.  Requires tv.cur < 0.
template <int MODE, class Stack, class Counters>
CRT_HD void trav_leaf(const SceneView &sc, f3 o, f3 d, float tmin, Search &s, int after_rank, float t_hi, Trav &tv, Stack &stack, Counters &cnt)
{
    const uint32_t bits = (uint32_t)~tv.cur;
    const uint32_t first = bits >> 3, count = (bits & 7u) + 1u;
    for (uint32_t k = 0; k < count; k++) {
        const q4 *tp = sc.tris + 3u * (first + k);
        const q4 ta = tp[0], tb = tp[1], tc = tp[2];
        float t;
        const int rank_word = f2i(tb.w);
        if (Counters::sphere_slots && (rank_word & CRT_SPHERE_SLOT)) { // only in scenes whose spheres went into the tree
            cnt.sphere();
            bool doubt = false;
            if (sphere_t<MODE != MODE_CLOSEST>(mk3(ta.x, ta.y, ta.z), tc.x, tb.x, tb.z, o, d, tmin, t, doubt)) {
                offer<MODE>(s, t, (int)(first + k) | CRT_SPHERE_SLOT, f2i(tc.w), rank_word & ~CRT_SPHERE_SLOT, after_rank, t_hi);
                if (MODE == MODE_CLOSEST) s.near_tie = doubt ? 1 : s.near_tie;
                if (MODE == MODE_CLOSEST) tv.hi = prune_limit<Counters::sphere_slots>(sc, tv, s.t) - tv.shift + CRT_TIE_WINDOW;
            }
            continue;
        }
        cnt.tri();
        if (triangle_t(mk3(ta.x, ta.y, ta.z), mk3(tb.x, tb.y, tb.z), mk3(tc.x, tc.y, tc.z), o, d, tmin, t)) {
            offer<MODE>(s, t, (int)(first + k), f2i(tc.w), rank_word, after_rank, t_hi);
            if (MODE == MODE_CLOSEST) tv.hi = prune_limit<Counters::sphere_slots>(sc, tv, s.t) - tv.shift + CRT_TIE_WINDOW;
        }
    }
    tv.cur = stack.pop();
}

// One whole walk over spheres + BVH.  WIDE: -1 = as the scene says (host code), 0 / 1 = the caller knows (the kernels are compiled
// for one node format each).
template <int MODE, class Stack, class Counters, int WIDE = -1>
CRT_HD void walk(const SceneView &sc, f3 o, f3 d, float tmin, Search &s, int after_rank, float t_hi, Stack &stack, Counters &cnt, Trav &tv)
{
    trav_begin<MODE>(sc, o, d, tmin, s, after_rank, t_hi, tv, stack, cnt);
    const bool wide = WIDE < 0 ? sc.wide != 0 : WIDE != 0;
    while (tv.cur != stack_none<Stack>::value) {
        if (wide) { while (is_inner<stack_none<Stack>::value>(tv.cur)) trav_inner4(sc, tv, stack, cnt); }
        else { while (is_inner<stack_none<Stack>::value>(tv.cur)) trav_inner(sc, tv, stack, cnt); }
        if (tv.cur < 0) trav_leaf<MODE>(sc, o, d, tmin, s, after_rank, t_hi, tv, stack, cnt);
    }
}

template <int MODE, class Stack, class Counters>
CRT_HD void walk(const SceneView &sc, f3 o, f3 d, float tmin, Search &s, int after_rank, float t_hi, Stack &stack, Counters &cnt)
{
    Trav tv;
    walk<MODE>(sc, o, d, tmin, s, after_rank, t_hi, stack, cnt, tv);
}

// The fold in reference leaf order over the candidates with t <= t_min + window: one MODE_NEXT_RANK walk per candidate,
// the reference's own comparisons.  Only what happens after m -- the first candidate in rank order that attains t_min --
// decides the result (m is accepted whenever it is visited and resets closest to t_min).  From there closest can creep up
// by at most 1e-7 per accepted triangle (triangle.h:75), so a chain of candidates spaced ~1e-7 apart can carry it to the
// end of the window and beyond; `reach` is the largest t the reference could still have accepted after m, and when it lies
// outside the window the fold is done again over a window that contains it.  (Nearly always one pass; a second one needs
// >= 9 successively accepted candidates behind the minimum.)
// `n` and `tv` are working storage (the render kernel hands in its lane's own search and walk state, which are dead while a
// hit is being resolved: the fold then needs no registers beyond the shade phase's and nothing is parked in scratch
// memory -- whose reload, a vector memory load, put an s_waitcnt vmcnt(0) on the COMMON path behind the pixel store);
// tv.cur is CRT_BVH_NONE again on return.
template <class Stack, class Counters, int WIDE = -1>
CRT_HD int refold(const SceneView &sc, f3 o, f3 d, float tmin, float tmax, float t_min, float &t_hit, int &mat_hit, Stack &stack, Counters &cnt, Search &n,
                  Trav &tv)
{
    // Provisional sphere hits (crt_core.h: sphere_t) come here too, with or without a tie.  Every walk below applies the reference's
    // literal box test to sphere hits, so a provisional hit that the test turns away is simply not among the candidates.  If it
    // had supplied t_min itself (it also pruned the common path's walk then), the valid candidates lie anywhere above t_min: those
    // inside the window are folded as usual -- t_min is only used as a lower bound of `closest` -- and if the window holds none the
    // fold is repeated WITHOUT a window, which is the reference's own walk (every primitive whose test succeeds, in rank order,
    // no distance culling).  That costs a walk per candidate along the whole ray and happens about once in 1e9 rays.
    float t_hi = t_min + CRT_TIE_WINDOW;
    float closest = tmax;
    int win = -1;
    // The very first walk looks for the LAST candidate of the window in rank order, L.  Whenever the reference visits L its
    // `closest` is the t of some candidate, i.e. >= t_min, so L is certainly accepted if it would be against t_min itself; and
    // if nothing behind L in rank order can be accepted after it (everything the window holds comes before L, and what a
    // triangle behind it could reach, t_L + 1e-7, is inside the window) L is the reference's answer -- without visiting the
    // other candidates.  That is the common tie (a shared edge, a doubled face: equal t), and it keeps k primitives that
    // coincide exactly from costing k walks of k leaves each.  Otherwise: the fold proper, below.  (One call site for both
    // kinds of walk: the walk is inlined, and the kernel holds this code in every variant.)
    int order = -1;
    for (int pass = 0; pass < 64; pass++) {
        closest = tmax;
        win = -1;
        int after = -1;
        bool seen_min = false;
        float reach = -FLT_MAX;
        for (;;) {
            n.t = 0.0f; n.code = -1; n.mat = 0; n.rank = 0x7FFFFFFF; n.near_tie = order;
            walk<MODE_NEXT_RANK, Stack, Counters, WIDE>(sc, o, d, tmin, n, order ? (int)0x80000000 : after, t_hi, stack, cnt, tv);
            if (order) {
                order = 0;
                const bool is_tri = n.code >= 0 && (uint32_t)n.code < sc.ntris;
                if (n.code >= 0 && (is_tri ? triangle_accepts(n.t, t_min) : sphere_accepts(n.t, t_min)) && !(n.t + 1e-7f > t_hi)) {
                    closest = n.t; win = n.code; mat_hit = n.mat;
                    break;
                }
                continue;
            }
            if (n.code < 0) break;
            const bool is_tri = (uint32_t)n.code < sc.ntris;
            seen_min = seen_min || n.t == t_min;
            if (is_tri ? triangle_accepts(n.t, closest) : sphere_accepts(n.t, closest)) {
                closest = n.t; win = n.code; mat_hit = n.mat;
                if (seen_min) reach = __builtin_fmaxf(reach, closest + 1e-7f); // what a later triangle may still reach (spheres need t <= closest)
            }
            after = n.rank;
        }
        if (win < 0 && t_hi < FLT_MAX) { t_hi = FLT_MAX; continue; } // t_min was a provisional hit that the box test turned away
        if (!(reach > t_hi)) break;
        t_hi = reach + CRT_TIE_WINDOW;
    }
    t_hit = closest;
    return win;
}
template <class Stack, class Counters>
CRT_HD int refold(const SceneView &sc, f3 o, f3 d, float tmin, float tmax, float t_min, float &t_hit, int &mat_hit, Stack &stack, Counters &cnt)
{
    Search n;
    Trav tv;
    return refold(sc, o, d, tmin, tmax, t_min, t_hit, mat_hit, stack, cnt, n, tv);
}

// After a MODE_CLOSEST walk: the reference's winner.  Fast path when no two candidates came within the tie
// window; otherwise the fold in reference leaf order.
template <class Stack, class Counters>
CRT_HD int resolve_hit(const SceneView &sc, f3 o, f3 d, float tmin, float tmax, const Search &s, float &t_hit, int &mat_hit, Stack &stack, Counters &cnt)
{
    if (s.code < 0) return -1;
    // sphere.h:36 / triangle.h:75 also bound t by the caller's tmax (FLT_MAX everywhere in the reference)
    if (!s.near_tie) { t_hit = s.t; mat_hit = s.mat; return s.t <= tmax ? s.code : -1; }
    cnt.refold();
    return refold(sc, o, d, tmin, tmax, s.t, t_hit, mat_hit, stack, cnt);
}
// The same for a caller whose search `s` and walk state `tv` are free to be used as the fold's working storage (see refold):
// `s` is consumed, tv.cur must be and stays CRT_BVH_NONE.
template <class Stack, class Counters, int WIDE = -1>
CRT_HD int resolve_hit_in_place(const SceneView &sc, f3 o, f3 d, float tmin, float tmax, Search &s, float &t_hit, int &mat_hit, Stack &stack, Counters &cnt,
                                Trav &tv)
{
    if (s.code < 0) return -1;
    if (!s.near_tie) { t_hit = s.t; mat_hit = s.mat; return s.t <= tmax ? s.code : -1; }
    cnt.refold();
    const float t_min = s.t;
    return refold<Stack, Counters, WIDE>(sc, o, d, tmin, tmax, t_min, t_hit, mat_hit, stack, cnt, s, tv);
}

// world->hit(ray, Interval(tmin, FLT_MAX), rec) (camera.h:68 -> hittable_list.h:17-63): returns the winning
// primitive code (-1 = miss) and its t.
template <class Stack, class Counters>
CRT_HD int closest_hit(const SceneView &sc, f3 o, f3 d, float tmin, float tmax, float &t_hit, int &mat_hit, Stack &stack, Counters &cnt)
{
    cnt.ray();
    Search s;
    s.t = tmax; s.code = -1; s.mat = 0; s.rank = 0; s.near_tie = false;
    walk<MODE_CLOSEST>(sc, o, d, tmin, s, 0, 0.0f, stack, cnt);
    return resolve_hit(sc, o, d, tmin, tmax, s, t_hit, mat_hit, stack, cnt);
}

// Hit record of the winning primitive: point, face-forward normal, primitive id (its material id travels with the search)
// (triangle.h:78-83, sphere.h:43-47, hittable.h:18-21).
struct Surface { f3 point, normal; bool front_face; int prim; };

// triangle.h:81 for the triangle of one 48-byte slot; w carries the primitive id
CRT_HD q4 triangle_normal(q4 ta, q4 tb, q4 tc)
{
    const f3 a = mk3(ta.x, ta.y, ta.z);
    const f3 n = normalize(cross(mk3(tb.x, tb.y, tb.z) - a, mk3(tc.x, tc.y, tc.z) - a));
    q4 r;
    r.x = n.x; r.y = n.y; r.z = n.z; r.w = ta.w;
    return r;
}

template <bool SPHERE_SLOTS = true>
CRT_HD Surface surface_at(const SceneView &sc, int code, f3 o, f3 d, float t)
{
    Surface r;
    r.point = axpy(t, d, o);
    if ((uint32_t)code < sc.ntris) {
        // normalize(cross(b - a, c - a)) (triangle.h:81) does not depend on the ray: evaluated once per triangle at commit by
        // triangle_normal() below -- the same operations, so the same bits -- and fetched as one 16-byte record instead of the
        // triangle's three
        const q4 nn = load_global(sc.normals, (size_t)code);
        r.normal = face_normal(d, mk3(nn.x, nn.y, nn.z), r.front_face);
        r.prim = f2i(nn.w);
    } else if (SPHERE_SLOTS && (code & CRT_SPHERE_SLOT)) {
        const size_t k = 3 * (size_t)(code & ~CRT_SPHERE_SLOT);
        const q4 ta = load_global(sc.tris, k), tb = load_global(sc.tris, k + 1);
        const f3 outward = div3_exact(r.point - mk3(ta.x, ta.y, ta.z), sphere_radius(tb.x), tb.y); // sphere.h:45 (the clamped radius is in the third quad; one comparison here is cheaper than its fetch)
        r.normal = face_normal(d, outward, r.front_face);
        r.prim = f2i(ta.w);
    } else {
        const Sphere sp = load_global(sc.spheres, (uint32_t)code - sc.ntris);
        const f3 outward = div3_exact(r.point - mk3(sp.cx, sp.cy, sp.cz), sp.radius, sp.inv_radius);
        r.normal = face_normal(d, outward, r.front_face);
        r.prim = sp.id;
    }
    return r;
}

// One iteration of Camera::ray_color's loop body after world->hit() (camera.h:68-83): `code`/`t` are the hit (code < 0:
// miss).  Updates colour / attenuation / ray; returns false when the path ends (miss, no scatter).
struct Path { f3 o, d, attenuation, color; };

template <class Counters>
CRT_HD bool path_bounce(const SceneView &sc, const CameraPOD &cam, Path &p, int code, float t, int mat, uint32_t depth, uint32_t pixel, uint32_t sample,
                        uint32_t k0, uint32_t k1, Counters &cnt, int &primary_prim)
{
    if (code < 0) {
        p.color = mk3(fma1(p.attenuation.x, cam.background.x, p.color.x), fma1(p.attenuation.y, cam.background.y, p.color.y),
                      fma1(p.attenuation.z, cam.background.z, p.color.z));
        return false;
    }
    const Shade m = sc.shades[mat];
    const Surface sf = surface_at(sc, code, p.o, p.d, t);
    cnt.shade();
    if (depth == 0) primary_prim = sf.prim;
    const f3 colour = texture_value(m, sf.point);
    const f3 e = emitted(m.type, colour);
    p.color = mk3(fma1(p.attenuation.x, e.x, p.color.x), fma1(p.attenuation.y, e.y, p.color.y), fma1(p.attenuation.z, e.z, p.color.z));
    if (!scatters(m.type)) return false; // material.h:14-16
    const u4 rnd = philox4x32_10(pixel, sample, 1u + depth, 0u, k0, k1);
    f3 att, no, nd;
    if (!scatter(m.type, m.param, colour, p.d, sf.point, sf.normal, sf.front_face, rnd, att, no, nd)) return false;
    p.attenuation = p.attenuation * att;
    p.o = no; p.d = nd;
    return true;
}

// Camera::ray_color (camera.h:60-87) for sample `sample` of pixel (i, j); primary_prim receives the id of
// the first hit (-1 = miss).  Sequential form (host single-stepper; the kernel schedules the same steps per lane).
template <class Stack, class Counters>
CRT_HD f3 sample_radiance(const SceneView &sc, const CameraPOD &cam, uint32_t i, uint32_t j, uint32_t pixel, uint32_t sample,
                          uint32_t max_depth, uint32_t k0, uint32_t k1, Stack &stack, Counters &cnt, int &primary_prim)
{
    Path p;
    camera_ray(cam, i, j, pixel, sample, k0, k1, p.o, p.d);
    p.attenuation = mk3(1.0f, 1.0f, 1.0f);
    p.color = mk3(0.0f, 0.0f, 0.0f);
    primary_prim = -1;
    for (uint32_t depth = 0; depth < max_depth; depth++) {
        float t = 0.0f;
        int mat = 0;
        const int code = closest_hit(sc, p.o, p.d, 0.001f, FLT_MAX, t, mat, stack, cnt);
        if (!path_bounce(sc, cam, p, code, t, mat, depth, pixel, sample, k0, k1, cnt, primary_prim)) break;
    }
    return p.color;
}

} // namespace crt
