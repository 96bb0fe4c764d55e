"""CPU tests of the oracle (oracle/crt_ref.c) itself: known-answer vectors, hand-computed cases per reference
function, and the oracle-of-the-oracle (brute-force fold in reference leaf order, hittable_list.h:22-30).

PARITY UNPINNED: the reference ships no tests or golden vectors and cannot be built here, so these
tests pin the restatement against independent arithmetic (Random123 Philox vectors, libm, hand-worked geometry), not
against outputs of the reference itself.
"""
import math
import os

import numpy as np
import pytest

from tests import helpers

FLT_MAX = 3.4028234663852886e38


def test_philox_known_answers(oracle):
    # Random123 kat_vectors for philox4x32-10 (also SURVEY.md section 4)
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_uniform_range_is_half_open_like_curand(oracle):
    L = oracle.lib()
    assert L.ref_u01(0) == pytest.approx(2.0 ** -24, abs=0) and L.ref_u01(0) > 0.0  # never 0 (utility.h:71-73, F14)
    assert L.ref_u01(0xFFFFFFFF) == 1.0                                              # 1 is reachable
    assert L.ref_u01(0x80000000) == 0.5 + 2.0 ** -24


def test_sincos_polynomial_against_libm(oracle):
    # the arguments utility.h:91-92 can produce: u * pi and u * 2pi for u on the 2^-24 grid
    rng = np.random.default_rng(1)
    u = (rng.integers(0, 2 ** 24, size=20000).astype(np.float64) + 1) * 2.0 ** -24
    xs = np.concatenate([(u.astype(np.float32) * np.float32(3.14159274101257324)), (u.astype(np.float32) * np.float32(6.28318548202514648)),
                         np.array([2.0 ** -24 * 3.1415927, 3.14159274, 6.2831855, 1.5707964, 4.712389], dtype=np.float32)]).astype(np.float32)
    worst = 0.0
    for x in xs:
        s, c = oracle.sincos(x)
        ts, tc = math.sin(float(x)), math.cos(float(x))
        # error in ulps of the true value, but never finer than 2^-24 absolute (near the zeros of sin/cos)
        for got, true in ((s, ts), (c, tc)):
            ulp = max(float(np.spacing(np.float32(abs(true)))), 2.0 ** -24)
            worst = max(worst, abs(got - true) / ulp)
    assert worst <= 2.0, worst


def test_camera_height_quirk_table(oracle):
    # camera.h:27: height = uint32(width / (16/9)) whatever `aspect` says (F12); SURVEY.md section 3.4 table
    for width, height in ((400, 225), (1000, 562), (1280, 720), (1920, 1080), (3840, 2160)):
        cam = oracle.RefCamera(width, (8, 5, -6), (-4, 0, 4), (0, 1, 0), 25.0, 1.0, 0.6, math.sqrt(125.0))
        hdr = np.frombuffer(cam.buf.raw[:16], dtype=np.uint32)
        assert (int(hdr[0]), int(hdr[1])) == (width, height)
        f = np.frombuffer(cam.buf.raw[8:16], dtype=np.float32)
        assert f[0] == np.float32(0.6)  # member default survives the shadowing parameter (camera.h:20,55)


def test_camera_ray_goes_through_focus_plane(oracle):
    cam = oracle.RefCamera(400, (8, 5, -6), (-4, 0, 4), (0, 1, 0), 25.0, 16 / 9, 0.6, math.sqrt(125.0))
    o, d = cam.get_ray(400, 200, 112, 0, 1984)
    # origin on the defocus disk (radius = focus * tan(0.3 deg)), direction unnormalised, length ~ focus distance (F11)
    assert np.linalg.norm(o - np.array([8, 5, -6], np.float32)) <= math.sqrt(125.0) * math.tan(math.radians(0.3)) * 1.0001
    assert abs(np.linalg.norm(d) - math.sqrt(125.0)) < 0.2
    o2, d2 = cam.get_ray(400, 200, 112, 0, 1984)
    assert np.array_equal(o, o2) and np.array_equal(d, d2)           # counter-based: same (pixel, sample) -> same ray
    o3, _ = cam.get_ray(400, 200, 112, 1, 1984)
    assert not np.array_equal(o, o3)


def test_camera_stage_stream_layout(oracle):
    """Stage 0 of the Philox stream: 16-bit uniforms, two per word, low half first; word 0 = square offsets, word 1 + k =
    defocus-disk draw k.  The ray is recomputed here from raw Philox words and the camera record."""
    W = 400
    cam = oracle.RefCamera(W, (8, 5, -6), (-4, 0, 4), (0, 1, 0), 25.0, 16 / 9, 0.6, math.sqrt(125.0))
    f = np.frombuffer(cam.buf.raw, dtype=np.float32)
    center, pixel00, pdu, pdv = f[4:7], f[7:10], f[10:13], f[13:16]
    disk_u, disk_v = f[25:28], f[28:31]
    f32 = np.float32

    def u16(h):
        return f32((h & 0xFFFF) + 1) * f32(2.0 ** -16)

    def fma(a, b, c):  # exact in double for these magnitudes, rounded once
        return f32(np.float64(a) * np.float64(b) + np.float64(c))

    needed_second_block = 0
    for (i, j, sample) in [(200, 112, 0), (0, 0, 5), (399, 224, 63), (17, 93, 2), (250, 10, 7), (3, 200, 1), (123, 45, 11), (77, 77, 9)]:
        pixel = j * W + i
        words = []
        for block in range(3):
            words += oracle.philox([pixel, sample, 0, block], [1984, 0])
        ox, oy = fma(u16(words[0]), f32(1.0), f32(-0.5)), fma(u16(words[0] >> 16), f32(1.0), f32(-0.5))
        for k in range(11):
            px, py = fma(u16(words[1 + k]), f32(2.0), f32(-1.0)), fma(u16(words[1 + k] >> 16), f32(2.0), f32(-1.0))
            if k == 10 or np.sqrt(fma(py, py, f32(px * px))) < 1.0:
                break
        needed_second_block += k >= 3
        o, d = cam.get_ray(W, i, j, sample, 1984)
        want_o = np.array([fma(py, disk_v[a], fma(px, disk_u[a], center[a])) for a in range(3)], f32)
        pc = np.array([fma(f32(oy + f32(j)), pdv[a], fma(f32(ox + f32(i)), pdu[a], pixel00[a])) for a in range(3)], f32)
        assert np.array_equal(o, want_o) and np.array_equal(d, (pc - want_o).astype(f32))
    assert needed_second_block <= 1


def test_slab_test_ignores_interval_and_handles_zero_direction(oracle):
    # aabb.h:72-92
    assert oracle.slab_test((0, 0, 0), (1, 1, 1), (-1, 0.5, 0.5), (1, 0, 0))        # inf * finite slabs pass
    assert not oracle.slab_test((0, 0, 0), (1, 1, 1), (-1, 2.0, 0.5), (1, 0, 0))    # outside the y slab, parallel
    assert not oracle.slab_test((0, 0, 0), (1, 1, 1), (2, 0.5, 0.5), (1, 0, 0))     # box behind the origin: tmax <= 0
    assert oracle.slab_test((0, 0, 0), (1, 1, 1), (0.5, 0.5, 0.5), (1, 2, 3))       # origin inside
    # 0 * inf = NaN when the origin lies exactly on a slab plane of a zero direction component.  With glm's
    # min(a,b) = (b<a)?b:a / max(a,b) = (a<b)?b:a the NaN stays in that axis' tNear/tFar but is dropped by the
    # reductions (NaN is the second operand there), so the axis simply does not constrain: the test passes.
    assert oracle.slab_test((0, 0, 0), (1, 1, 1), (-1, 0.0, 0.5), (1, 0, 0))
    assert not oracle.slab_test((0, 0, 0), (1, 1, 1), (-1, 0.0, 1.5), (1, 0, 0))


def test_slab_test_nan_case_depends_on_the_axis(oracle):
    """aabb.h:73-91 with a direction component of +-0 and the origin EXACTLY on one of that axis' box planes: (plane - o) *
    (1 / 0) = 0 * inf = NaN for that plane.  glm::min(a, b) = (b < a) ? b : a and glm::max(a, b) = (a < b) ? b : a return
    their FIRST operand when either is NaN, so in max(max(tNear.x, tNear.y), tNear.z) a NaN in y or z is dropped (second
    operand) while a NaN in x is the first operand of both reductions and survives: tmin = NaN, `tmin <= tmax` is false, the
    reference MISSES the box although the ray runs along its face; on the max plane the NaN turns tFar into -inf and every
    axis misses.  This KAT records what the restatement does; the product's
    pruning arithmetic (crt_traverse.h: reciprocal clamped to +-1e30, so 0 * 1e30 = 0) visits such a box, i.e. this one case is
    outside the equivalence argument of DESIGN.md section 2 -- it needs a path segment whose direction has an exactly zero
    component AND whose origin equals, in that component and bit for bit, a 1e-4-padded box plane of the reference tree."""
    box = ((0, 0, 0), (1, 1, 1))
    for zero in (0.0, -0.0):
        # x axis: origin.x on the min plane (t0.x = NaN: first operand of glm::min -> tNear.x = NaN) or on the max plane
        # (t1.x = NaN: glm::min(t0, NaN) = t0 = -inf/+inf side, glm::max(t0, NaN) = t0): only the first poisons tmin
        assert not oracle.slab_test(*box, (0.0, -1.0, 0.5), (zero, 1.0, 0.0))      # NaN survives -> miss (the reference's quirk)
        # y / z axes: the NaN is a second operand of the reductions and is dropped -> the other axes decide -> hit
        assert oracle.slab_test(*box, (-1.0, 0.0, 0.5), (1.0, zero, 0.0))
        assert oracle.slab_test(*box, (-1.0, 0.5, 0.0), (1.0, 0.0, zero))
        # origin on the MAX plane of the zero axis: t1 = NaN, t0 = -inf, glm::max(-inf, NaN) = -inf -> tFar = -inf on that axis
        # -> tmax = -inf: a miss whatever the axis
        assert not oracle.slab_test(*box, (1.0, -1.0, 0.5), (zero, 1.0, 0.0))
        assert not oracle.slab_test(*box, (-1.0, 1.0, 0.5), (1.0, zero, 0.0))
        assert not oracle.slab_test(*box, (-1.0, 0.5, 1.0), (1.0, 0.0, zero))
    # a zero component with the origin strictly inside / outside the slab: +-inf, no NaN, the ordinary answers
    assert oracle.slab_test(*box, (0.5, -1.0, 0.5), (0.0, 1.0, 0.0))
    assert not oracle.slab_test(*box, (1.5, -1.0, 0.5), (0.0, 1.0, 0.0))


def test_triangle_hit_cases(oracle):
    a, b, c = (0, 0, 0), (1, 0, 0), (0, 1, 0)
    t, n, front = oracle.triangle_hit(a, b, c, (0.25, 0.25, 1), (0, 0, -1))
    assert t == 1.0 and n == (0.0, 0.0, 1.0) and front is True
    t, n, front = oracle.triangle_hit(a, b, c, (0.25, 0.25, -2), (0, 0, 4))          # back face, unnormalised direction (F11)
    assert t == 0.5 and n == (0.0, 0.0, -1.0) and front is False
    assert oracle.triangle_hit(a, b, c, (0.25, 0.25, 1), (1, 0, 0)) is None         # parallel: |det| < 1e-7
    assert oracle.triangle_hit(a, b, c, (0.75, 0.75, 1), (0, 0, -1)) is None        # u + v > 1
    assert oracle.triangle_hit(a, b, c, (0.0, 0.0, 1), (0, 0, -1)) is not None      # vertex hit, inside the 1e-7 band
    assert oracle.triangle_hit(a, b, c, (0.5, 0.5, 1), (0, 0, -1)) is not None      # edge hit
    assert oracle.triangle_hit(a, b, c, (0.25, 0.25, 1), (0, 0, -1), tmin=0.001, tmax=0.5) is None  # beyond closest_so_far
    # triangle.h:75: a hit up to 1e-7 beyond closest_so_far is still accepted
    assert oracle.triangle_hit(a, b, c, (0.25, 0.25, 0.75), (0, 0, -1), tmax=np.float32(0.75) - np.float32(5e-8)) is not None


def test_sphere_hit_cases(oracle):
    t, n, front = oracle.sphere_hit((0, 0, 0), 1.0, (0, 0, 3), (0, 0, -1))
    assert t == 2.0 and n == (0.0, 0.0, 1.0) and front
    t, n, front = oracle.sphere_hit((0, 0, 0), 1.0, (0, 0, 0), (0, 0, -2))           # from inside: far root, flipped normal
    assert t == 0.5 and n == (0.0, 0.0, 1.0) and not front
    assert oracle.sphere_hit((0, 0, 0), 1.0, (0, 2, 3), (0, 0, -1)) is None
    t, _, _ = oracle.sphere_hit((0, 0, 0), 1.0, (0, 1, 3), (0, 0, -1))               # tangent: discriminant 0
    assert t == 3.0
    assert oracle.sphere_hit((0, 0, 0), 1.0, (0, 0, 3), (0, 0, -1), tmax=2.0) is not None   # closed interval (interval.h:19-21)
    assert oracle.sphere_hit((0, 0, 0), 1.0, (0, 0, 3), (0, 0, -1), tmax=1.9999) is None    # near root beyond -> far root beyond


def test_morton_code(oracle):
    L = oracle.lib()
    assert L.ref_morton_3d(0.0, 0.0, 0.0) == 0
    assert L.ref_morton_3d(1.0, 1.0, 1.0) == 0x3FFFFFFF                                   # clamped to 1023 per axis
    assert L.ref_morton_3d(1.0 / 1024, 0.0, 0.0) == 4 and L.ref_morton_3d(0.0, 1.0 / 1024, 0.0) == 2 and L.ref_morton_3d(0.0, 0.0, 1.0 / 1024) == 1
    assert L.ref_morton_3d(-5.0, 0.5, 2.0) == L.ref_morton_3d(0.0, 0.5, 1.0)


def test_model_statistics(scenes):
    cat = scenes.cat()
    assert cat["meshes"][0]["indices"].shape == (732, 3)                                  # 732 / 2196 triangles
    v = cat["meshes"][0]["vertices"]
    assert np.allclose(v.min(0), [-1.731696, 0.097411, -1.565736], atol=1e-6) and np.allclose(v.max(0), [1.19502, 2.119426, 1.086331], atol=1e-6)
    cats = scenes.cats()
    assert sum(m["indices"].shape[0] for m in cats["meshes"]) == 2196


def test_reference_bvh_shape(oracle, scenes):
    rs = oracle.RefScene(scenes.cat())
    assert rs.size == 734
    codes, idx = rs.mortons()
    assert np.all(np.diff(codes.astype(np.int64)) >= 0) and sorted(idx.tolist()) == list(range(734))
    # F10: the radius-1000 ground sphere dominates the scene box, so the cat collapses onto a handful of Morton codes
    assert len(np.unique(codes)) < 40
    assert np.allclose(rs.bbox(), [-1000, -2000, -1001, 1000, 8.5, 999])
    # stable sort: equal codes keep primitive order
    for k in range(733):
        if codes[k] == codes[k + 1]:
            assert idx[k] < idx[k + 1]
    assert np.array_equal(rs.ranks()[idx], np.arange(734))


@pytest.mark.parametrize("name,width,height", [("cat", 400, 400), ("cats", 320, 180)])
def test_bvh_traversal_equals_brute_force_in_leaf_order(oracle, scenes, name, width, height):
    """The reference's candidate set is hierarchy independent: its LBVH walk (no distance culling, F8) must equal a fold
    over ALL primitives in DFS-leaf (= sorted Morton) order.  This is what lets the product use another BVH."""
    desc = getattr(scenes, name)()
    rs = oracle.RefScene(desc)
    params, _ = scenes.make_camera(width)
    cam = oracle.make_camera(params, width)
    fb0, ids0, _, st0 = rs.render(cam, width, height, 2, mode=oracle.TRAVERSE_BVH)
    fb1, ids1, _, st1 = rs.render(cam, width, height, 2, mode=oracle.TRAVERSE_BRUTE_LEAF_ORDER)
    assert np.array_equal(ids0, ids1) and np.array_equal(fb0, fb1) and st0.rays == st1.rays
    assert (ids0 >= 0).any() and (ids0 < rs.size - 2).any()


def test_list_order_fold_differs_only_at_ties(oracle, scenes):
    # folding in list order instead of leaf order may only change results where hits tie within 1e-7 (F9)
    desc = scenes.cat()
    rs = oracle.RefScene(desc)
    params, _ = scenes.make_camera(400)
    cam = oracle.make_camera(params, 400)
    _, ids0, _, _ = rs.render(cam, 400, 400, 1, mode=oracle.TRAVERSE_BVH)
    _, ids2, _, _ = rs.render(cam, 400, 400, 1, mode=oracle.TRAVERSE_BRUTE_LIST_ORDER)
    assert (ids0 != ids2).mean() < 1e-3


def test_coincident_triangles_last_leaf_wins(oracle):
    # two identical triangles: the one later in leaf order wins the tie (hittable_list.h:47-51, triangle.h:75)
    from cudaraytracer_amd import capi
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    desc = dict(textures=[], materials=[dict(type=capi.MAT_LAMBERTIAN, rgb=(0.5, 0.5, 0.5), param=0.0, texture=None)],
                meshes=[dict(vertices=v, indices=np.array([[0, 1, 2], [0, 1, 2], [0, 1, 2]], np.uint32), material=0)],
                spheres=[dict(center=(0, -1000, 0), radius=1.0, material=0)])
    rs = oracle.RefScene(desc)
    ranks = rs.ranks()
    prim, t = rs.trace((0.25, 0.25, 1), (0, 0, -1))
    assert t == 1.0 and prim == int(np.argmax(ranks[:3]))


def test_golden_fixtures(oracle, scenes):
    """Regression fixtures written by tests/golden/make_golden.py from this oracle (not from the reference: none exist)."""
    g = np.load(os.path.join(helpers.GOLDEN, "cat_400x400_primary_ids.npz"))
    rs = oracle.RefScene(scenes.cat())
    params, _ = scenes.make_camera(400)
    cam = oracle.make_camera(params, 400)
    fb, ids, _, st = rs.render(cam, 400, 400, 16)
    assert np.array_equal(ids, g["ids"].astype(np.int32))
    assert st.rays == int(g["rays_16spp"])
    assert np.array_equal(fb.reshape(50, 8, 50, 8, 3).mean(axis=(1, 3)).astype(np.float32), g["tile_means_16spp"]) or \
        np.allclose(fb.reshape(50, 8, 50, 8, 3).mean(axis=(1, 3)), g["tile_means_16spp"], rtol=0, atol=1e-6)
    import hashlib
    assert hashlib.sha256(fb.tobytes()).hexdigest() == str(g["sha256_16spp"])


def test_length_thresholds_without_square_root():
    """crt_core.h compares lengths with 1 and 1e-8 on the squared length (length_lt_1, length_le_1em8): the float
    neighbourhoods of both thresholds agree with the reference's sqrt form (camera.h:111-119 via utility.h:100-109,
    material.h:29)."""
    def around(x, n=200000):
        bits = np.array([x], dtype=np.float32).view(np.uint32)[0].astype(np.int64)
        return (bits + np.arange(-n, n + 1)).astype(np.uint32).view(np.float32)
    one = around(1.0)
    assert np.array_equal(np.sqrt(one) < np.float32(1.0), one < np.float32(1.0))
    thr = np.float32(float.fromhex("0x1.cd2b2ap-54"))
    tiny = around(thr)
    assert np.array_equal(np.sqrt(tiny) <= np.float32(1e-8), tiny <= thr)
    coarse = np.float32(10.0) ** np.arange(-44, 38, dtype=np.float32)   # and far from them
    assert np.array_equal(np.sqrt(coarse) <= np.float32(1e-8), coarse <= thr)
    assert np.array_equal(np.sqrt(coarse) < np.float32(1.0), coarse < np.float32(1.0))
