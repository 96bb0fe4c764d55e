// crt_core.h -- per-lane arithmetic of the path-tracing hot path (gfx950 device code; also compilable for the
// host so tests can single-step it without a GPU -- see tests/hostsim/).
//
// Every float operation that reaches an output is spelled out here (explicit __builtin_fmaf, no compiler
// contraction: build with -ffp-contract=off, IEEE-correct division and sqrt) so that the result is a pure
// function of the inputs, independent of compiler and of lane/wave scheduling.  Reference citations are to
// /root/reference/CudaRayTracer/.
#pragma once

#include <stdint.h>
#include <float.h>
#include <math.h>

#if defined(__HIPCC__)
#define CRT_HD __host__ __device__ __forceinline__
#else
#define CRT_HD inline
#endif

namespace crt {

// ---------------------------------------------------------------------------------------------------------
// vec3 (stands in for glm::vec3; glm is not used anywhere)
// ---------------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };

CRT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
CRT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
CRT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
CRT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
CRT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
CRT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
CRT_HD f3 div3(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
CRT_HD float fma1(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// a + s*b, one fma per component
CRT_HD f3 axpy(float s, f3 b, f3 a) { return mk3(fma1(s, b.x, a.x), fma1(s, b.y, a.y), fma1(s, b.z, a.z)); }

// a / s for a divisor whose correctly rounded reciprocal y = RN(1 / s) came with the scene: q = a * y, then one correction by
// the exact remainder, q + fma(-s, q, a) * y -- three instructions per component where the IEEE division sequence takes
// thirteen.  That this IS the correctly rounded quotient is not taken on trust: for every divisor that gets a reciprocal,
// commit has compared it with a / s for all 2^23 significands of a (crt_bvh.cpp: reciprocal_if_exact; scaling a by a
// power of two scales every intermediate exactly as long as nothing leaves the normal range, hence the range check
// here and the divisor range there).  y == 0: no reciprocal, plain division.
CRT_HD f3 div3_exact(f3 a, float s, float y)
{
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    if (y != 0.0f && lo >= 0x1p-60f && hi <= 0x1p+60f) {
        const float qx = a.x * y, qy = a.y * y, qz = a.z * y;
        return mk3(fma1(fma1(-s, qx, a.x), y, qx), fma1(fma1(-s, qy, a.y), y, qy), fma1(fma1(-s, qz, a.z), y, qz));
    }
    return div3(a, s);
}
// glm::dot order (x*x' + y*y') + z*z', contracted left to right
CRT_HD float dot(f3 a, f3 b) { return fma1(a.z, b.z, fma1(a.y, b.y, a.x * b.x)); }
// glm::cross, each difference of products as fma(p, q, -(r*s))
CRT_HD f3 cross(f3 a, f3 b)
{
    return mk3(fma1(a.y, b.z, -(b.y * a.z)), fma1(a.z, b.x, -(b.z * a.x)), fma1(a.x, b.y, -(b.x * a.y)));
}
CRT_HD float length(f3 a) { return __builtin_sqrtf(dot(a, a)); }
CRT_HD f3 normalize(f3 a) { return a * (1.0f / __builtin_sqrtf(dot(a, a))); } // glm: v * inversesqrt(dot(v,v))
// Comparisons of a length with a constant, without the square root.  sqrtf is correctly rounded and monotonic, so
// sqrtf(x) < 1 <=> x < 1 (the root of the float below 1 rounds below 1) and sqrtf(x) <= 1e-8f <=> x <= CRT_SQ_1EM8, the
// largest float whose root rounds to <= 1e-8f (tests/test_oracle.py walks the floats around both thresholds).
#define CRT_SQ_1EM8 0x1.cd2b2ap-54f
CRT_HD bool length_lt_1(float len2) { return len2 < 1.0f; }
CRT_HD bool length_le_1em8(float len2) { return len2 <= CRT_SQ_1EM8; }

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10, counter = (pixel, sample, stage, block), key = seed.  Replaces curand XORWOW
// (renderer.h:26,45; utility.h:71-73) so that a (pixel, sample, bounce) triple names its random numbers.
// stage 0 = camera ray, stage 1 + depth = scatter at bounce `depth`.
// ---------------------------------------------------------------------------------------------------------
struct u4 { uint32_t x, y, z, w; };

CRT_HD uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

CRT_HD u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#if defined(__clang__)
#pragma unroll
#endif
    for (int r = 0; r < 10; r++) {
        // both halves of each 32 x 32 product from ONE 64-bit multiply: gfx950's v_mad_u64_u32 issues at the rate of a single
        // v_mul_hi_u32 (4.6 cycles per wave and SIMD, tools/ubench/valu_rate2.hip), so the round's four multiplies become two
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}
// (0,1] on a 2^-24 grid: same range as curand_uniform (utility.h:71-73), exactly representable.
CRT_HD float u01(uint32_t x) { return (float)((x >> 8) + 1u) * 5.9604644775390625e-08f; }

// ---------------------------------------------------------------------------------------------------------
// sin / cos for arguments in (0, 2*pi] (the only ones utility.h:91-95 produces): quadrant reduction with a
// three-part pi/2 and degree-7 / degree-8 polynomials, every step an explicit fma.  <= 1.5 ulp.
// ---------------------------------------------------------------------------------------------------------
CRT_HD void sincos_02pi(float x, float &s, float &c)
{
    const float kf = __builtin_rintf(x * 0.636619772367581343f);
    const int k = (int)kf;
    float r = fma1(kf, -1.5707962512969971f, x);
    r = fma1(kf, -7.5497894158615964e-08f, r);
    r = fma1(kf, -5.3903029534742384e-15f, r);
    const float r2 = r * r;
    float ps = fma1(r2, 2.6083159809786593541503e-06f, -1.981069071916863322258e-04f);
    ps = fma1(ps, r2, 8.333307858556509017944336e-03f);
    ps = fma1(ps, r2, -1.666666567325592041015625e-01f);
    const float sr = fma1(r * r2, ps, r);
    float pc = fma1(r2, 2.443315711809948e-05f, -1.388731625493765e-03f);
    pc = fma1(pc, r2, 4.166664568298827e-02f);
    pc = fma1(pc, r2, -0.5f);
    const float cr = fma1(r2, pc, 1.0f);
    const float ss = (k & 1) ? cr : sr;
    const float cc = (k & 1) ? sr : cr;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
}

#define CRT_PI_F 3.14159274101257324f  // (float)M_PI
#define CRT_2PI_F 6.28318548202514648f // (float)(2.0 * M_PI)

// utility.h:90-98 random_unit_vector: theta ~ U(0,pi], phi ~ U(0,2pi] -- not uniform on the sphere, kept (F13).
// u_theta, u_phi are the two uniforms in draw order.
CRT_HD f3 random_unit_vector(float u_theta, float u_phi)
{
    const float theta = fma1(u_theta, CRT_PI_F - 0.0f, 0.0f);
    const float phi = fma1(u_phi, CRT_2PI_F - 0.0f, 0.0f);
    float st, ct, sp, cp;
    sincos_02pi(theta, st, ct);
    sincos_02pi(phi, sp, cp);
    return mk3(st * cp, st * sp, ct);
}

// utility.h:111-113
CRT_HD f3 reflect(f3 v, f3 n) { const float k = 2.0f * dot(v, n); return axpy(-k, n, v); }

// ---------------------------------------------------------------------------------------------------------
// Intersection arithmetic.  Both tests return the reference's t for the primitive *as a function of the ray
// and tmin only*; the caller folds candidates against closest_so_far (see crt_traverse.h for why that is
// equivalent to hittable_list.h:47-51).
// ---------------------------------------------------------------------------------------------------------
// triangle.h:48-76 (Moller-Trumbore).  true -> t is a candidate with t >= tmin - 1e-7.
CRT_HD bool triangle_t(f3 a, f3 b, f3 c, f3 o, f3 d, float tmin, float &t_out)
{
    const f3 edge_ab = b - a;
    const f3 edge_ac = c - a;
    const f3 pvec = cross(d, edge_ac);
    const float det = dot(edge_ab, pvec);
    const float epsilon = 1e-7f;
    if (__builtin_fabsf(det) < epsilon) return false;
    const float inv_det = 1.0f / det;
    const f3 tvec = o - a;
    const float u = dot(tvec, pvec) * inv_det;
    if (u < -epsilon || u > 1.0f + epsilon) return false;
    const f3 qvec = cross(tvec, edge_ab);
    const float v = dot(d, qvec) * inv_det;
    if (v < -epsilon || u + v > 1.0f + epsilon) return false;
    const float t = dot(edge_ac, qvec) * inv_det;
    if (t < tmin - epsilon) return false;
    t_out = t;
    return true;
}
// triangle.h:75: the upper half of the interval test, against the running closest_so_far
CRT_HD bool triangle_accepts(float t, float closest) { return !(t > closest + 1e-7f); }

// aabb.h:72-92 fastAABBIntersect, literally (glm::min(a, b) = (b < a) ? b : a, glm::max(a, b) = (a < b) ? b : a, so a NaN
// from 0 * inf travels the way it does there).  The walk's own box tests only prune and never use this; it is evaluated for
// the one box whose verdict the sphere test does not imply (below).
CRT_HD float glm_min(float a, float b) { return (b < a) ? b : a; }
CRT_HD float glm_max(float a, float b) { return (a < b) ? b : a; }
// (Not inlined on the device: it only runs on the rare careful path, and inlined there its temporaries raise the register
// pressure of the whole kernel -- scratch spills on the common path, +5 % frame time.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline)) bool reference_box_test(f3 bmin, f3 bmax, f3 o, f3 d)
#else
inline bool reference_box_test(f3 bmin, f3 bmax, f3 o, f3 d)
#endif
{
    const float eps = 1e-7f;
    const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    const float t0x = (bmin.x - o.x) * ix, t0y = (bmin.y - o.y) * iy, t0z = (bmin.z - o.z) * iz;
    const float t1x = (bmax.x - o.x) * ix, t1y = (bmax.y - o.y) * iy, t1z = (bmax.z - o.z) * iz;
    const float nx = glm_min(t0x, t1x) - eps, ny = glm_min(t0y, t1y) - eps, nz = glm_min(t0z, t1z) - eps;
    const float fx = glm_max(t0x, t1x) + eps, fy = glm_max(t0y, t1y) + eps, fz = glm_max(t0z, t1z) + eps;
    const float t_near = glm_max(glm_max(nx, ny), nz);
    const float t_far = glm_min(glm_min(fx, fy), fz);
    return t_near <= t_far && t_far > 0.0f;
}

// sphere.h:22-41.  The reference tries the near root, then the far root, each against [tmin, closest]; since
// far >= near, the root it can return is the first one >= tmin, accepted iff <= closest (sphere_accepts).
// `radius_arg` is the constructor's argument: the sphere's radius is max(0, radius_arg) (sphere.h:11), its leaf box
// center -+ radius_arg, ordered per axis (sphere.h:12-15, aabb.h:12-16) and NOT padded.
// The reference only reaches Sphere::hit through the boxes of the sphere's leaf and of the leaf's ancestors (bvh.h), and for
// a sphere its own test does not imply theirs: the discriminant b*b - a*c cancels (unnormalised directions, |d| ~ 14 for a
// camera ray), so a ray that passes ~1e-4 OUTSIDE the silhouette can come out of it as a hit, and where the silhouette
// touches the box -- the six axis-extreme points -- such a ray misses the box and the reference never tests the sphere.
// (Found by the parity fuzz, tools/fuzz_parity.py: about one ray in 1e9 of a sphere scene.)  So a sphere hit also has to
// pass the reference's test of the sphere's own box, in the reference's arithmetic.  The ancestors' boxes need no test:
// they contain the leaf box, and every operation of the slab test is monotone in the box coordinates (subtract, multiply by
// a fixed reciprocal, min / max, -+ eps), so -- NaN cases aside, DESIGN.md section 2 -- a ray that passes the test of the
// leaf box passes theirs.  Triangles need none of this: their leaf boxes are padded by 1e-4 (triangle.h), two orders of
// magnitude above the slack (1e-7 in u, v) and the rounding (~1e-6 at these scene sizes) of Triangle::hit.
//
// The literal test costs three IEEE divisions and ~60 further operations: +6.6 % of the headline frame when every sphere hit pays
// it (the ground sphere is a candidate of nearly every ray), and still +5 % when it merely sits in a branch of the common path
// (register pressure).  So the common path (STRICT = false) never evaluates it: sphere_hit_implies_box() decides nearly all hits
// from what Sphere::hit has already computed, and a hit it cannot vouch for (1 in ~4000 on the cats frame) is taken PROVISIONALLY
// and reported in `doubt`: the search then goes the careful way the ties go (crt_traverse.h: refold), which repeats the nearest-hit
// walk with STRICT = true -- the literal test for every sphere hit.
// The proof, with rounding bounds (u = 2^-24, q = |oc|^2, G = |c|inf + r): the EXACT ray passes through the ball of radius
// r - u G (which lies inside the float-rounded box) ahead of the origin, over a parameter interval long enough for the float slab
// test to see it, whenever  disc' >= u a W,  b' > 0  and  c' >= 0  (the primes: as computed), with W = 28 q + r (24 r + 8 |c|inf):
//   * |disc' - disc| <= u a (19 q + 6 r^2) for the operations of Sphere::hit (oc, the three dots, fma(-r, r, q), a * c,
//     fma(b, b, -ac));
//   * shrinking the ball by u G costs at most 2 a r u G of discriminant;
//   * every plane time of the slab test carries a relative error <= 3u (subtract, reciprocal, multiply), so near' <= far' is
//     certain once the interval is >= 12u times its larger end, i.e. disc >= 37 u^2 b^2 -- far below the two terms above;
//   * the interval ends at a positive time (the reference also asks t_far > 0) if the centre lies ahead, b > 0: with c' >= 0 the
//     first condition gives b'^2 >= 27 u a q, i.e. |b'| >= 5e-4 sqrt(a q), against |b' - b| <= 4u sqrt(a q).
// W covers all of these with room to spare (and for the rounding of the bound itself); its ray-independent part is evaluated at
// commit (sphere_sure_term, kept in the sphere's record), which leaves one fma, one multiplication and three comparisons per hit.
// A ray that starts INSIDE the sphere or leaves its surface (c' < 0 or b' <= 0: the far root, 0 +- noise for a leaving ray, can
// exceed tmin) gets a second chance, as does everything else the first test cannot vouch for: if the origin lies strictly between
// the planes as the reference rounds them, then in every axis the two plane distances have opposite signs, their products with
// 1/d keep those signs exactly, and t_near' <= -eps < eps <= t_far' whatever the direction.  (It matters for the reference scene
// itself: the ground sphere touches its box right under the models, where short rays leave the ground.)
// What neither proves (a band ~ 1e-3 |oc|^2 / r wide along the silhouette -- on the cats frame the rays that graze the horizon,
// ~2e-5 of all -- NaN, r = 0 seen from outside the box) is a doubt.  tools/sphere_box_audit.py: no hit the shortcut vouches for
// fails the literal test in 1e8 rays made to be hard for it (and a bound 50x smaller fails thousands).
CRT_HD float sphere_sure_term(f3 center, float radius)
{
    const float cmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(center.x), __builtin_fabsf(center.y)), __builtin_fabsf(center.z));
    return radius * fma1(8.0f, cmax, 24.0f * radius) * 5.9604644775390625e-08f;
}
CRT_HD bool sphere_hit_implies_box(float sure, float a, float b, float q, float c, float discriminant)
{
    return discriminant >= a * fma1(28.0f * 5.9604644775390625e-08f, q, sure) && b > 0.0f && c >= 0.0f;
}
CRT_HD bool strictly_inside_box(f3 center, float h, f3 o)
{
    return center.x - h < o.x && o.x < center.x + h && center.y - h < o.y && o.y < center.y + h && center.z - h < o.z && o.z < center.z + h;
}

template <bool STRICT>
CRT_HD bool sphere_t(f3 center, float radius, float radius_arg, float sure, f3 o, f3 d, float tmin, float &t_out, bool &doubt)
{
    // radius = sphere_radius(radius_arg) = std::max(0.0f, radius_arg) (sphere.h:11), evaluated at commit and kept in the record
    const f3 oc = center - o;
    const float a = dot(d, d);
    const float b = dot(d, oc);
    const float q = dot(oc, oc);
    const float c = fma1(-radius, radius, q);
    const float discriminant = fma1(b, b, -(a * c));
    if (discriminant < 0) return false;
    const float sd = __builtin_sqrtf(discriminant);
    float root = (b - sd) / a;
    if (!(root >= tmin)) {
        root = (b + sd) / a;
        if (!(root >= tmin)) return false;
    }
    const float h = __builtin_fabsf(radius_arg); // center - r < center + r for r > 0, the other way round for r < 0: the same box
    if (STRICT) {
        if (!reference_box_test(mk3(center.x - h, center.y - h, center.z - h), mk3(center.x + h, center.y + h, center.z + h), o, d)) return false;
    } else {
        doubt = !sphere_hit_implies_box(sure, a, b, q, c, discriminant);
        if (doubt) doubt = !strictly_inside_box(center, h, o);
    }
    t_out = root;
    return true;
}
CRT_HD bool sphere_accepts(float t, float closest) { return t <= closest; }

// ---------------------------------------------------------------------------------------------------------
// Records shared by host builder and kernels
// ---------------------------------------------------------------------------------------------------------
struct Material { int type; float r, g, b; float param; int tex; int pad0, pad1; };               // 32 B
struct Texture { int type; float ar, ag, ab; float br, bg, bb; float inv_scale; };                // 32 B
struct Sphere { float cx, cy, cz, radius_arg; int id; int rank; int mat; float inv_radius; float sure; float radius; int pad1, pad2; }; // 48 B; radius_arg: the constructor's argument, radius = max(0, radius_arg) (sphere.h:11); inv_radius: RN(1 / radius) if commit proved div3_exact with it, else 0; sure: sphere_sure_term()
CRT_HD float sphere_radius(float radius_arg) { return (0.0f < radius_arg) ? radius_arg : 0.0f; }
// Device record of a material joined with its texture (48 B, built at commit): one fetch per shaded hit instead
// of material -> texture.  Colour a/b: the texture of a Lambertian / DiffuseLight (texture.h), the albedo of a Metal
// (material.h:48) and (1,1,1) for a Dielectric (material.h:66), so that every material's colour is texture_value().
struct alignas(16) Shade {
    int type; float param; int tex_type; float inv_scale;
    float ar, ag, ab, pad1; // solid colour / odd checker cells
    float br, bg, bb, pad2; // even checker cells
};

enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIFFUSE_LIGHT = 2, MAT_DIELECTRIC = 3 };
enum { TEX_SOLID = 0, TEX_CHECKER = 1 };

struct CameraPOD { // mirrors crt_camera (include/crt.h)
    uint32_t image_width, image_height;
    float defocus_angle, focus_distance;
    f3 center, pixel00, pdu, pdv, u, v, w, disk_u, disk_v, background;
};

// texture.h:16, 31-39.  `a` is returned where the floor-sum is odd (the reference's `is_even` is true then).
CRT_HD f3 texture_value(const Shade &t, f3 p)
{
    const f3 ca = mk3(t.ar, t.ag, t.ab);
    if (t.tex_type == TEX_SOLID) return ca;
    const int xi = (int)__builtin_floorf(t.inv_scale * p.x);
    const int yi = (int)__builtin_floorf(t.inv_scale * p.y);
    const int zi = (int)__builtin_floorf(t.inv_scale * p.z);
    const bool pick_a = ((xi + yi + zi) % 2) != 0;
    return pick_a ? ca : mk3(t.br, t.bg, t.bb);
}

// hittable.h:18-21
CRT_HD f3 face_normal(f3 d, f3 outward, bool &front_face)
{
    front_face = dot(d, outward) < 0.0f;
    return front_face ? outward : -outward;
}

// utility.h:115-120
CRT_HD f3 refract(f3 uv, f3 n, float etai_over_etat)
{
    const float cos_theta = __builtin_fminf(dot(-uv, n), 1.0f);
    const f3 r_out_perp = axpy(cos_theta, n, uv) * etai_over_etat;
    const float l = length(r_out_perp);
    const f3 r_out_parallel = n * (-__builtin_sqrtf(__builtin_fabsf(fma1(-l, l, 1.0f))));
    return r_out_perp + r_out_parallel;
}
// material.h:96-101, pow(x, 5) as ((x*x)*(x*x))*x
CRT_HD float reflectance(float cosine, float ri)
{
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    const float x = 1.0f - cosine, x2 = x * x;
    return fma1(1.0f - r0, (x2 * x2) * x, r0);
}

// material.h scatter().  rnd = first Philox block of stage 1+depth (x, y = theta, phi uniforms; the dielectric
// uses x).  Returns false when the path ends (base Material / DiffuseLight, material.h:14-16).
// `colour` = texture_value() of the material's Shade record at the hit point.
CRT_HD bool scatter(int type, float param, f3 colour, f3 ray_d, f3 point, f3 normal, bool front_face, u4 rnd,
                    f3 &attenuation, f3 &new_o, f3 &new_d)
{
    f3 no = point, nd = ray_d; // locals, written out once (keeps the outputs in registers)
    bool ok = false;
    if (type == MAT_LAMBERTIAN || type == MAT_METAL) {
        const f3 ruv = random_unit_vector(u01(rnd.x), u01(rnd.y)); // shared by both: the wave evaluates it once
        if (type == MAT_LAMBERTIAN) { // material.h:24-34
            f3 dir = normal + ruv;
            if (length_le_1em8(dot(dir, dir))) dir = normal; // length(dir) <= 1e-8f
            nd = dir;
        } else { // material.h:44-51 (always scatters)
            const f3 reflected = reflect(ray_d, normal);
            nd = axpy(param, ruv, normalize(reflected));
        }
        ok = true;
    } else if (type == MAT_DIELECTRIC) { // material.h:61-91
        const float ri = front_face ? 1.0f / param : param;
        const f3 unit_direction = normalize(ray_d);
        const float cos_theta = __builtin_fminf(dot(-unit_direction, normal), 1.0f);
        const float sin_theta = __builtin_sqrtf(fma1(-cos_theta, cos_theta, 1.0f));
        const bool is_reflect = ri * sin_theta > 1.0f;
        f3 direction;
        if (is_reflect || reflectance(cos_theta, ri) > fma1(u01(rnd.x), 1.0f - 0.0f, 0.0f)) direction = reflect(unit_direction, normal);
        else direction = refract(unit_direction, normal, ri);
        no = axpy(1e-3f, direction, point);
        nd = direction;
        ok = true;
    }
    attenuation = colour;
    new_o = no; new_d = nd;
    return ok;
}
CRT_HD bool scatters(int type) { return type == MAT_LAMBERTIAN || type == MAT_METAL || type == MAT_DIELECTRIC; }
// material.h:10-12, 109-111
CRT_HD f3 emitted(int type, f3 colour) { return type == MAT_DIFFUSE_LIGHT ? colour : mk3(0.0f, 0.0f, 0.0f); }

// (0,1] on a 2^-16 grid from the low half of x: the camera stage's uniforms, two per Philox word (low half first).
CRT_HD float u01_16(uint32_t x) { return (float)((x & 0xFFFFu) + 1u) * 1.52587890625e-05f; }

// camera.h:88-98 get_ray + 111-119.  Philox stage 0, 16-bit uniforms: block 0 = word 0 (square.x, square.y), words 1..3 =
// disk draws 0..2; block 1 = draws 3..6; block 2 = draws 7..10.  utility.h:100-109 makes up to 10 rejection rounds and
// returns the 11th draw unconditionally.  A second block is needed by 0.9 % of the rays (0.2146^3).
// `r` = Philox block 0 of stage 0 (the caller may have evaluated it together with other lanes' scatter blocks).
CRT_HD void camera_ray_from(const CameraPOD &cam, uint32_t i, uint32_t j, uint32_t pixel, uint32_t sample, uint32_t k0, uint32_t k1, u4 r,
                            f3 &o, f3 &d)
{
    const float ox = fma1(u01_16(r.x), 0.5f - -0.5f, -0.5f);
    const float oy = fma1(u01_16(r.x >> 16), 0.5f - -0.5f, -0.5f);
    const f3 pixel_center = axpy(oy + (float)j, cam.pdv, axpy(ox + (float)i, cam.pdu, cam.pixel00));
    f3 origin = cam.center;
    if (!(cam.defocus_angle <= 0.0f)) {
        uint32_t w = r.y; // draw k lives in word 1 + k of the stage's stream
        float px = fma1(u01_16(w), 1.0f - -1.0f, -1.0f);
        float py = fma1(u01_16(w >> 16), 1.0f - -1.0f, -1.0f);
        uint32_t draw = 0;
        // draws 0..9 are tested; draw 10 is taken as is
        while (draw < 10 && !length_lt_1(fma1(0.0f, 0.0f, fma1(py, py, px * px)))) { // glm::length(p) < 1
            draw++;
            const uint32_t word = draw + 1u;
            if ((word & 3u) == 0u) r = philox4x32_10(pixel, sample, 0u, word >> 2, k0, k1);
            w = (word & 2u) ? ((word & 1u) ? r.w : r.z) : ((word & 1u) ? r.y : r.x);
            px = fma1(u01_16(w), 2.0f, -1.0f); py = fma1(u01_16(w >> 16), 2.0f, -1.0f);
        }
        origin = axpy(py, cam.disk_v, axpy(px, cam.disk_u, cam.center));
    }
    o = origin;
    d = pixel_center - origin;
}
CRT_HD void camera_ray(const CameraPOD &cam, uint32_t i, uint32_t j, uint32_t pixel, uint32_t sample, uint32_t k0, uint32_t k1,
                       f3 &o, f3 &d)
{
    camera_ray_from(cam, i, j, pixel, sample, k0, k1, philox4x32_10(pixel, sample, 0u, 0u, k0, k1), o, d);
}

} // namespace crt
