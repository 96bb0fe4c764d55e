"""CPU tests of the product's host side: the C ABI surface (include/crt.h), scene preparation (reference ranks,
SAH BVH), the camera, error behaviour -- and the product's per-lane traversal / tie-fold / shading code compiled for
the host (tests/hostsim, test-only) against the oracle.  No kernel runs here; rendering without a GPU must fail loudly.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests import helpers

REPO = helpers.REPO


def test_library_exports_every_declared_symbol(capi):
    header = open(os.path.join(REPO, "include", "crt.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(crt_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(capi.SYMBOLS)
    L = C.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert capi.lib().crt_version() == 200


def test_struct_layouts_match_header(capi):
    assert C.sizeof(capi.Camera) == 136 and C.sizeof(capi.RenderParams) == 56 and C.sizeof(capi.Stats) == 136
    assert C.sizeof(capi.SceneInfo) == 104  # (96 until round 3: sah_cost appended, padded to the struct's 8-byte alignment)


def test_render_without_device_fails_loudly(capi, scenes):
    s = scenes.build(scenes.cat(), capi.DEVICE_NONE)
    _, cam = scenes.make_camera(64)
    with pytest.raises(capi.CrtError) as e:
        s.render(cam, 64, 36, 1)
    assert e.value.code == capi.CRT_ERR_NO_DEVICE


def test_argument_errors(capi):
    s = capi.Scene()
    with pytest.raises(capi.CrtError):
        s.commit(capi.DEVICE_NONE)                      # empty scene
    with pytest.raises(capi.CrtError):
        s.add_sphere((0, 0, 0), 1.0, 0)                 # no such material
    m = s.add_material(capi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))
    with pytest.raises(capi.CrtError):
        s.add_mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 3]], np.uint32), m)   # index out of range (scene.h:134-136)
    with pytest.raises(capi.CrtError):
        s.add_material(capi.MAT_LAMBERTIAN, (1, 1, 1), 0.0, 99)                           # no such texture
    with pytest.raises(capi.CrtError):
        s.add_material(7, (1, 1, 1))
    with pytest.raises(capi.CrtError) as e:
        capi.load_obj("/nonexistent/file.obj")
    assert e.value.code == capi.CRT_ERR_IO


def test_obj_loader_variants(capi, tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("# quad with negative indices, v//vn and a polygon\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\n"
                 "f -4//1 -3//1 -2//1 -1//1\nf 1 2 3\n")
    v, i = capi.load_obj(str(p))
    assert i.shape == (3, 3)                                 # quad fan-triangulated + one triangle
    assert v.shape == (7, 6)                                 # 4 joined (v,vn) corners + 3 without normals
    assert np.array_equal(v[i[0], :3], [[0, 0, 0], [1, 0, 0], [1, 1, 0]]) and np.array_equal(v[i[1], :3], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    assert np.array_equal(v[0, 3:], [0, 0, 1]) and np.array_equal(v[4, 3:], [0, 0, 0])


@pytest.mark.parametrize("name", ["cat", "cats", "cat_checker"])
def test_reference_ranks_and_camera_match_oracle(capi, scenes, oracle, name):
    desc = getattr(scenes, name)()
    s = scenes.build(desc, capi.DEVICE_NONE)
    rs = oracle.RefScene(desc)
    assert np.array_equal(s.ranks(), rs.ranks())            # bvh.h:13-48 + scene.h:299-301, computed independently on both sides
    for width in (400, 1920, 3840):
        params, cam = scenes.make_camera(width)
        assert bytes(cam) == oracle.make_camera(params, width).buf.raw


def test_bvh_shape(capi, scenes):
    for name, ntri in (("cat", 732), ("cats", 2196)):
        info = scenes.build(getattr(scenes, name)(), capi.DEVICE_NONE).info()
        assert info.num_triangles == ntri and info.num_spheres == 2
        assert 1 <= info.max_leaf_size <= 4 and info.bvh_depth <= 32
        assert info.node_bytes == 32 * info.node_width * info.num_nodes and info.triangle_bytes == 48 * ntri
        assert ntri / 4 <= info.num_nodes + 1 <= ntri
        # root box = triangles only: the radius-1000 ground sphere does not inflate the tree (F10)
        assert max(abs(x) for x in list(info.bounds_min) + list(info.bounds_max)) < 10


def test_depth_bound_on_degenerate_input(capi):
    # 5000 nested, strongly overlapping slivers: SAH alone would build a very deep tree; the builder must stay <= 32
    n = 5000
    k = np.arange(n, dtype=np.float32)
    v = np.zeros((3 * n, 3), np.float32)
    v[0::3] = np.stack([0 * k, 0 * k, 0 * k], 1)
    v[1::3] = np.stack([1.0001 ** k, 0 * k, 0 * k], 1)
    v[2::3] = np.stack([0 * k, 1.0001 ** k, 1e-3 * k], 1)
    s = capi.Scene()
    m = s.add_material(capi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))
    s.add_mesh(v, np.arange(3 * n, dtype=np.uint32).reshape(-1, 3), m)
    s.commit(capi.DEVICE_NONE)
    assert s.info().bvh_depth <= 32


@pytest.mark.parametrize("name,width,height,spp", [("cat", 200, 200, 4), ("cats", 320, 180, 2), ("cat_checker", 240, 135, 3)])
def test_single_stepped_product_code_is_bit_exact_with_oracle(hostsim, oracle, scenes, name, width, height, spp):
    desc = getattr(scenes, name)()
    s = hostsim.build(desc)
    params, cam = scenes.make_camera(width)
    rs = oracle.RefScene(desc)
    fb, ids, _, st = rs.render(oracle.make_camera(params, width), width, height, spp)
    out, oid, hs = hostsim.render(s, cam, width, height, spp)
    assert np.array_equal(ids, oid)                           # primary-hit primitive ids: bit exact
    assert np.array_equal(fb, out)                            # image: bit exact (same arithmetic, same RNG stream)
    assert st.rays == hs["rays"]                              # deterministic path-segment count
    assert hs["max_stack"] <= 32
    assert hs["nodes"] * 2 < st.box_tests                     # and the product BVH does far less work per ray


def test_member_defocus_angle_takes_the_pinhole_branch(hostsim, oracle, scenes):
    """camera.h:94: get_ray tests the MEMBER defocus_angle.  The constructor leaves it at 0.6 whatever its parameter says
    (F12); a caller that writes 0 (or less) to the public member afterwards gets rays from the camera centre and no disk
    draws.  Product code (single-stepped) against the oracle, and the ray itself against first principles."""
    desc = scenes.cat()
    s = hostsim.build(desc)
    rs = oracle.RefScene(desc)
    W, H = 96, 54
    for angle in (0.0, -2.0):
        params, cam = scenes.make_camera(W, member_defocus_angle=angle)
        assert cam.defocus_angle == angle
        rc = oracle.make_camera(params, W)
        o, d = rc.get_ray(W, 40, 20, 3, 1984)
        assert np.array_equal(o, np.array(params["lookfrom"], np.float32))       # the centre itself, bit for bit
        fb, ids, _, st = rs.render(rc, W, H, 3)
        out, oid, hs = hostsim.render(s, cam, W, H, 3)
        assert np.array_equal(ids, oid) and np.array_equal(fb, out) and st.rays == hs["rays"]
    # the branch changes the frame: with the member at its default the disk is sampled
    params, cam = scenes.make_camera(W)
    out_disk, _, _ = hostsim.render(s, cam, W, H, 3)
    assert not np.array_equal(out, out_disk)
    # a constructor parameter of 0 does NOT take the branch: radius 0, but the draws are made (same origin, other code path)
    params0, cam0 = scenes.make_camera(W, defocus_angle=0.0)
    assert cam0.defocus_angle == np.float32(0.6)
    o0, _ = oracle.make_camera(params0, W).get_ray(W, 40, 20, 3, 1984)
    assert np.array_equal(o0, np.array(params0["lookfrom"], np.float32))


def test_depth_parameter_and_sample_offset(hostsim, oracle, scenes):
    desc = scenes.cat()
    s = hostsim.build(desc)
    params, cam = scenes.make_camera(160)
    rs = oracle.RefScene(desc)
    rc = oracle.make_camera(params, 160)
    fb, ids, acc, _ = rs.render(rc, 160, 90, 2, max_depth=3, sample_offset=5, ids_sample=6, want_accum=True)
    out, oid, _ = hostsim.render(s, cam, 160, 90, 2, max_depth=3, sample_offset=5, ids_sample=6)
    assert np.array_equal(fb, out) and np.array_equal(ids, oid)
    out_acc, _, _ = hostsim.render(s, cam, 160, 90, 2, max_depth=3, sample_offset=5, ids_sample=6, flags=1)
    assert np.array_equal(acc, out_acc)


def test_forced_tie_path_agrees_with_fast_path(hostsim, oracle, scenes):
    desc = scenes.cats()
    s = hostsim.build(desc)
    params, _ = scenes.make_camera(192)
    rs = oracle.RefScene(desc)
    rc = oracle.make_camera(params, 192)
    n = 0
    for j in range(40, 108, 3):
        for i in range(0, 192, 3):
            o, d = rc.get_ray(192, i, j, 0, 1984)
            ref_prim, ref_t = rs.trace(o, d)
            p0, t0, _ = hostsim.trace(s, o, d)
            p1, t1, took = hostsim.trace(s, o, d, force_tie_path=True)
            assert p0 == ref_prim == p1 and (ref_prim < 0 or (t0 == ref_t == t1))
            n += took
    assert n > 0


def _tie_scene(capi):
    """Coincident and nearly coincident triangles + a sphere tangent to them: every hit is a tie decided by leaf order."""
    rng = np.random.default_rng(7)
    base = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    verts, idx = [], []
    for k in range(12):
        dz = np.float32(0.0 if k < 6 else (k - 5) * 4e-8)          # 6 exact duplicates, 6 within a few 1e-7 of t
        shift = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 0], np.float32) if k % 2 else np.zeros(3, np.float32)
        verts.append(base + shift + np.array([0, 0, dz], np.float32))
        idx.append([3 * k, 3 * k + 1, 3 * k + 2])
    mats = [dict(type=capi.MAT_LAMBERTIAN, rgb=(0.1 * (k + 1), 0.5, 0.5), param=0.0, texture=None) for k in range(3)]
    return dict(textures=[], materials=mats,
                meshes=[dict(vertices=np.concatenate(verts), indices=np.array(idx[:6], np.uint32), material=0),
                        dict(vertices=np.concatenate(verts), indices=np.array(idx[6:], np.uint32), material=1)],
                spheres=[dict(center=(0.0, 0.0, -1.0), radius=1.0, material=2), dict(center=(0.0, 0.0, -1.0), radius=1.0, material=1)])


def test_ties_resolve_in_reference_leaf_order(hostsim, oracle, capi):
    desc = _tie_scene(capi)
    s = hostsim.build(desc)
    rs = oracle.RefScene(desc)
    assert np.array_equal(s.ranks(), rs.ranks())
    rng = np.random.default_rng(3)
    took_any = 0
    for _ in range(400):
        o = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), 1.0], np.float32)
        d = np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), -1.0], np.float32)
        ref_prim, ref_t = rs.trace(o, d)
        prim, t, took = hostsim.trace(s, o, d)
        assert prim == ref_prim, (o, d)
        if ref_prim >= 0:
            assert t == ref_t
        took_any += took
    assert took_any > 100                                      # the scene really exercises the tie path
    # and as an image, through the whole path loop
    cam_params = dict(lookfrom=(0.0, 0.0, 4.0), lookat=(0.0, 0.0, 0.0), vup=(0.0, 1.0, 0.0), vfov=40.0, aspect=16 / 9, defocus_angle=0.6, focus_dist=4.0)
    cam = capi.camera(64, **{k: cam_params[k] for k in ("lookfrom", "lookat", "vup", "vfov", "aspect", "defocus_angle", "focus_dist")})
    rc = oracle.make_camera(cam_params, 64)
    fb, ids, _, st = rs.render(rc, 64, 36, 4)
    out, oid, hs = hostsim.render(s, cam, 64, 36, 4)
    assert np.array_equal(ids, oid) and np.array_equal(fb, out) and st.rays == hs["rays"] and hs["tie_refolds"] > 0


def _grazing_rays(desc, count, seed):
    """Rays tangent to a sphere of `desc` at (or up to ~1e-4 off) one of its six axis-extreme points -- where the sphere touches its own
    unpadded box -- with direction lengths 1, 14 (a camera ray's) and 40."""
    rng = np.random.default_rng(seed)
    sph = desc["spheres"]
    for _ in range(count):
        sp = sph[rng.integers(len(sph))]
        c = np.array(sp["center"], np.float64)
        r = abs(sp["radius"])
        ax = rng.integers(3)
        sg = rng.choice([-1.0, 1.0])
        e = c.copy()
        e[ax] += sg * r
        u = rng.normal(size=3)
        u[ax] = rng.normal() * 1e-3
        u /= np.linalg.norm(u)
        o = e - rng.uniform(5, 30) * u
        o[ax] += sg * rng.uniform(-1, 3) * 1e-4 * max(r, 1.0) * (0.05 if r < 10 else 1)
        yield o.astype(np.float32), (u * rng.choice([1.0, 14.0, 40.0])).astype(np.float32)


@pytest.mark.parametrize("keep", [488, 6])
def test_a_sphere_hit_outside_the_spheres_own_box_is_not_a_hit(hostsim, oracle, scenes, keep):
    """Sphere::hit's discriminant cancels, so a ray ~1e-4 outside the silhouette can come out as a hit; where the silhouette touches the
    sphere's unpadded leaf box the reference's box test (aabb.h:72-92) has already turned such a ray away (found by tools/fuzz_parity.py).
    The product applies that box test to every sphere hit (crt_core.h: sphere_t): it must agree with the oracle's literal BVH walk --
    which also visits the ancestors' boxes -- on rays made to graze there, for spheres in the tree (488) and in the flat list (6)."""
    desc = scenes.random_spheres()
    desc["spheres"] = desc["spheres"][:keep]
    s = hostsim.build(desc)
    rs = oracle.RefScene(desc)
    phantom = hits = 0
    for o, d in _grazing_rays(desc, 6000, keep):
        ref = rs.trace(o, d)
        prim, t, _ = hostsim.trace(s, o, d)
        assert prim == ref[0] and (prim < 0 or t == ref[1]), (o, d, prim, t, ref)
        hits += prim >= 0
        phantom += rs.trace(o, d, mode=oracle.TRAVERSE_BRUTE_LEAF_ORDER) != ref
    assert hits > 3000 and phantom >= 10                       # the rays really include hits that only the box test rejects


def test_negative_radius_argument(hostsim, oracle, scenes, capi):
    """sphere.h:11-13: the radius is max(0, argument) but the box is built from the argument itself.  A sphere given -0.5 is a point with
    a box of half-width 0.5: rays aimed at the point hit it or not as the discriminant's rounding decides, the same way on both sides."""
    desc = scenes._base()
    desc["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.5, 0.5, 0.5), param=0.0, texture=None))
    desc["spheres"] = [dict(center=(0.25, 0.5, -0.75), radius=-0.5, material=0), dict(center=(3.0, 0.5, 1.0), radius=0.5, material=0),
                       dict(center=(-2.0, 0.25, 0.5), radius=-0.0, material=0)]
    s = hostsim.build(desc)
    rs = oracle.RefScene(desc)
    rng = np.random.default_rng(11)
    point_hits = 0
    for k in range(3000):
        sp = desc["spheres"][k % 3]
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        L = rng.choice([1.0, 2.0, 4.0, 8.0])
        o = (np.array(sp["center"]) - L * u).astype(np.float32)
        d = (np.array(sp["center"], np.float32) - o) if k % 2 else u.astype(np.float32)
        ref = rs.trace(o, d)
        prim, t, _ = hostsim.trace(s, o, d)
        assert prim == ref[0] and (prim < 0 or t == ref[1]), (o, d, prim, t, ref)
        point_hits += prim in (0, 2)
    assert point_hits > 50


def _hovering_spheres(scenes, capi, hover):
    """15 spheres of radius 0.2 (more than 8: they go into the tree) `hover` above a two-triangle floor."""
    desc = scenes._base()
    desc["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.5, 0.5, 0.5), param=0.0, texture=None))
    centres = [(float(x), 0.2 + hover, float(z)) for x in range(-2, 3) for z in range(-1, 2)]
    desc["spheres"] = [dict(center=c, radius=0.2, material=0) for c in centres]
    v = np.array([[-50, 0, -50], [50, 0, -50], [50, 0, 50], [-50, 0, 50]], np.float32)
    desc["meshes"].append(dict(vertices=v, indices=np.array([[0, 1, 2], [0, 2, 3]], np.uint32), material=0))
    return desc, centres


def test_a_sphere_root_before_its_own_box_is_not_pruned(hostsim, oracle, scenes, capi):
    """A sphere's computed root carries its discriminant's cancellation and can lie before the entry time of the sphere's box.  Rays
    from BELOW a floor, aimed under spheres that hover 2e-4 above it: the reference (no distance culling) takes the sphere whenever
    its noisy root comes out smaller than the floor's t.  With the triangle walk's pruning limit (best + 1e-6) the product answered
    the floor for ~10 % of these rays; kernels for scenes with spheres in the tree prune at the root's error bound (crt_traverse.h:
    prune_limit; tools/sphere_prune_probe.py is the longer version of this test)."""
    desc, centres = _hovering_spheres(scenes, capi, 2e-4)
    s = hostsim.build(desc)
    rs = oracle.RefScene(desc)
    rng = np.random.default_rng(1)
    sphere_in_front_of_its_box = 0
    for _ in range(6000):
        c = np.array(centres[rng.integers(len(centres))])
        target = np.array([c[0], 0.0, c[2]]) + np.array([rng.normal(), 0, rng.normal()]) * 0.2 * 10.0 ** rng.uniform(-4, -1)
        az, el = rng.uniform(0, 2 * np.pi), 10.0 ** rng.uniform(-2, 0.19)
        u = np.array([np.cos(az) * np.cos(el), -np.sin(el), np.sin(az) * np.cos(el)])
        o = (target + 10.0 ** rng.uniform(0, 2) * u).astype(np.float32)
        d = (-u * 10.0 ** rng.uniform(-1, 1.3)).astype(np.float32)
        ref = rs.trace(o, d)
        prim, t, _ = hostsim.trace(s, o, d)
        assert prim == ref[0] and (prim < 0 or t == ref[1]), (o, d, prim, t, ref)
        sphere_in_front_of_its_box += 2 <= ref[0]            # prims 0, 1: the floor's triangles
    assert sphere_in_front_of_its_box > 200                   # (measured ~600: the reference answers a sphere that lies BEHIND the floor)


def test_fuzz_findings_round_3(hostsim, oracle, scenes):
    """The two frames tools/fuzz_parity.py found (one ray each whose sphere 'hit' lies outside the sphere's box), at their own parameters."""
    desc = scenes.random_spheres()
    s = hostsim.build(desc)
    rs = oracle.RefScene(desc)
    for w, h, extra, kw in ((191, 39, {}, dict(max_depth=4, seed=813596887249, sample_offset=45, ids_sample=45)),
                            (179, 27, dict(defocus_angle=0.0), dict(max_depth=7, seed=775945998277, sample_offset=33, ids_sample=33))):
        cam_kw = dict(scenes.SPHERES_CAMERA)
        cam_kw.update(extra)
        params, cam = scenes.make_camera(w, **cam_kw)
        spp = 1 if w == 191 else 8
        fb, ids, _, st = rs.render(oracle.make_camera(params, w), w, h, spp, **kw)
        out, oid, hs = hostsim.render(s, cam, w, h, spp, **kw)
        assert st.rays == hs["rays"] and np.array_equal(ids, oid) and np.array_equal(fb, out)


def _stacked_copies(scenes, capi, copies):
    """`copies` exact copies of each of three triangles (two of them sharing an edge), shuffled, + the reference spheres: every hit is
    a tie among `copies` (or, on the shared edge, 2 x copies) candidates with equal t."""
    d = scenes._base()
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.7, 0.3, 0.16), param=0.0, texture=None))
    scenes._add_ref_spheres(d)
    base = np.array([[[-1.5, 0.2, 0.0], [0.5, 0.2, 0.3], [-0.4, 2.2, -0.2]], [[0.5, 0.2, 0.3], [1.8, 0.3, -0.5], [-0.4, 2.2, -0.2]],
                     [[-1.0, 0.1, 1.2], [1.0, 0.1, 1.2], [0.0, 1.0, 1.0]]], np.float32)
    order = np.random.default_rng(copies).permutation(3 * copies) % 3
    v = base[order].reshape(-1, 3)
    d["meshes"].append(dict(vertices=np.ascontiguousarray(v), indices=np.arange(v.shape[0], dtype=np.uint32).reshape(-1, 3), material=0))
    return d


def test_exactly_coincident_primitives_take_one_fold_walk(hostsim, oracle, scenes, capi):
    """k primitives with equal t: the reference's answer is the last of them in leaf order.  The fold finds it with ONE walk
    for the largest rank (crt_traverse.h: refold) instead of k walks of k leaves; the frame is the oracle's either way, and the
    work counters say which way it went."""
    copies = 150
    desc = _stacked_copies(scenes, capi, copies)
    s = hostsim.build(desc)
    width, height, spp = 64, 36, 2
    params, cam = scenes.make_camera(width)
    fb, ids, _, st = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp)
    out, oid, hs = hostsim.render(s, cam, width, height, spp)
    assert np.array_equal(ids, oid) and np.array_equal(fb, out) and st.rays == hs["rays"]
    hit = (ids >= 0) & (ids < 3 * copies)
    assert hit.sum() > 50 and hs["tie_refolds"] >= hit.sum()
    print("coincident copies: %d folds, %d triangle tests, %d rays" % (hs["tie_refolds"], hs["tri_tests"], hs["rays"]))
    # per fold: the search that raised the flag + its repeat with literal box tests + one walk for the last rank, each testing the
    # ~150 (300 on the shared edge) copies under the ray once; the candidate-by-candidate fold would test them 150 times over
    assert hs["tri_tests"] < 10 * copies * hs["tie_refolds"]  # (measured 8.9; candidate by candidate it is ~150)


def test_four_child_nodes_on_the_host(hostsim, oracle, capi, scenes, monkeypatch):
    """CRT_BVH_WIDE=1 (crt_bvh.h: widen_bvh; crt_traverse.h: trav_inner4_from): the collapsed tree walked by the product code on the
    host gives the oracle's frames -- tie scene included, where the fold in leaf order decides every hit -- with about half the
    node fetches, at most one stack entry per level of the wide tree."""
    monkeypatch.setenv("CRT_BVH_WIDE", "1")
    for name, W, H, spp in (("cats", 160, 90, 2), ("random_spheres", 96, 54, 2)):
        desc = getattr(scenes, name)()
        s = hostsim.build(desc)
        info = s.info()
        assert info.node_width == 4 and info.node_bytes == 128 * info.num_nodes and info.bvh_depth < info.binary_depth
        params, cam = scenes.make_camera(W)
        fb, ids, _, st = oracle.RefScene(desc).render(oracle.make_camera(params, W), W, H, spp)
        out, oid, hs = hostsim.render(s, cam, W, H, spp)
        assert np.array_equal(ids, oid) and np.array_equal(fb, out) and st.rays == hs["rays"]
        assert hs["max_stack"] <= info.bvh_depth
        monkeypatch.setenv("CRT_BVH_WIDE", "0")
        two = hostsim.render(hostsim.build(desc), cam, W, H, spp)[2]
        monkeypatch.setenv("CRT_BVH_WIDE", "1")
        assert hs["nodes"] < 0.75 * two["nodes"] and hs["tri_tests"] <= 1.1 * two["tri_tests"]
    desc = _tie_scene(capi)
    s = hostsim.build(desc)
    assert s.info().node_width == 4
    rs = oracle.RefScene(desc)
    rng = np.random.default_rng(5)
    took_any = 0
    for _ in range(300):
        o = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), 1.0], np.float32)
        d = np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), -1.0], np.float32)
        ref_prim, ref_t = rs.trace(o, d)
        prim, t, took = hostsim.trace(s, o, d)
        assert prim == ref_prim and (ref_prim < 0 or t == ref_t), (o, d)
        took_any += took
    assert took_any > 50


def test_tile_partition_covers_frame_exactly_once(hostsim, scenes):
    desc = scenes.cat()
    s = hostsim.build(desc)
    _, cam = scenes.make_camera(100)
    full, ids_full, _ = hostsim.render(s, cam, 100, 61, 1)      # ragged: 100x61 is not a multiple of 8
    acc = np.zeros_like(full)
    cover = np.zeros((61, 100), np.int32)
    for r in range(3):
        part, ids, _ = hostsim.render(s, cam, 100, 61, 1, tile_rank=r, tile_ranks=3)
        cover += (ids != -2)
        acc += part
    assert np.all(cover == 1) and np.array_equal(acc, full)
    comp = np.concatenate([helpers.to_compact(full, r, 3) for r in range(3)])
    assert comp.size == 3 * capi_compact(100, 61, 3)
    assert np.array_equal(helpers.untile_numpy(comp, 100, 61, 3), full)


def capi_compact(w, h, r):
    from cudaraytracer_amd import capi
    return capi.compact_floats(w, h, r)


def test_sphere_scene_with_spheres_in_the_tree(hostsim, oracle, scenes):
    """Row N3 (scene.h:53-90): 488 spheres, Lambertian / Metal / Dielectric.  487 of them are sphere slots of the BVH (the
    r = 1000 ground stays in the flat list); ids, image and segment count must still be the oracle's."""
    desc = scenes.random_spheres()
    s = hostsim.build(desc)
    info = s.info()
    assert info.num_spheres == 488 and info.num_triangles == 0 and info.sphere_bytes == 48 and info.num_nodes > 200
    width, height, spp = 240, 135, 3
    params, cam = scenes.make_camera(width, **scenes.SPHERES_CAMERA)
    fb, ids, _, st = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp)
    out, oid, hs = hostsim.render(s, cam, width, height, spp)
    assert np.array_equal(ids, oid) and np.array_equal(fb, out) and st.rays == hs["rays"]
    assert len(np.unique(ids)) > 100 and fb.max() > 0.5          # the frame does show the spheres under a sky


def test_obj_groups_and_mtl_records(capi, tmp_path):
    """Row N4 (crt_load_obj_ex): objects / usemtl runs with their .mtl records; indices always refer to the one joined vertex
    array (the reference concatenates Assimp meshes without the base-vertex offset, model_loader.h:26-49)."""
    (tmp_path / "two.mtl").write_text(
        "newmtl red\nKd 0.9 0.1 0.1\nNs 10\n\nnewmtl lamp\nKd 0 0 0\nKe 4 5 6\nillum 2\nNi 1.45\nd 0.5\n")
    (tmp_path / "two.obj").write_text(
        "mtllib two.mtl\n"
        "o floor\nv 0 0 0\nv 1 0 0\nv 1 0 1\nv 0 0 1\nusemtl red\nf 1 2 3 4\n"
        "o bulb\nv 0 1 0\nv 1 1 0\nv 0 1 1\nusemtl lamp\nf 5 6 7\nusemtl nowhere\nf -3 -2 -1\nf 1 5 6\n")
    v, i, tg, groups = capi.load_obj_groups(str(tmp_path / "two.obj"))
    assert v.shape == (7, 6) and i.shape == (5, 3)
    assert [g["object"] for g in groups] == ["floor", "bulb", "bulb"] and [g["material"] for g in groups] == ["red", "lamp", "nowhere"]
    assert [(g["first_triangle"], g["num_triangles"]) for g in groups] == [(0, 2), (2, 1), (3, 2)] and tg.tolist() == [0, 0, 1, 2, 2]
    assert np.allclose(groups[0]["Kd"], (0.9, 0.1, 0.1)) and groups[0]["Ns"] == 10.0 and groups[0]["illum"] == 1
    assert np.allclose(groups[1]["Ke"], (4, 5, 6)) and groups[1]["illum"] == 2 and abs(groups[1]["Ni"] - 1.45) < 1e-6 and groups[1]["d"] == 0.5
    assert np.allclose(groups[2]["Kd"], (0.8, 0.8, 0.8)) and np.allclose(groups[2]["Ke"], 0)      # unknown material: MTL defaults
    assert i[3].tolist() == [4, 5, 6] and i[4].tolist() == [0, 4, 5]                                # relative and cross-object indices
    # the shipped models: one object, one material each (Blender's lambert1, Kd 0.8)
    _, idx, tg, groups = capi.load_obj_groups(os.path.join(helpers.REPO, "models", "cats.obj"))
    assert idx.shape == (2196, 3) and len(groups) == 1 and groups[0]["material"] == "lambert1" and np.allclose(groups[0]["Kd"], 0.8)
    v0, i0 = capi.load_obj(os.path.join(helpers.REPO, "models", "cats.obj"))
    assert np.array_equal(i0, idx)


def _creeping_chain_scene(capi, n=16, step=8.9e-8):
    """n parallel triangles stacked `step` apart along +z, seen from z < 0.  The reference's leaf order is the Morton order of
    the centroids, ascending in z among otherwise equal cells (ties: insertion order), i.e. nearest first: every triangle is
    within 1e-7 of the one before it and
    is accepted in turn (triangle.h:75), closest_so_far creeps from the nearest to the farthest, (n - 1) x 1e-7 behind the
    minimum -- beyond the 1e-6 window the tie fold starts with.  (8.9e-8 = 2.99 ulp of t ~ 0.25: consecutive t differ by at
    most 3 ulp, which fl(closest + 1e-7f) always covers.)"""
    base = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    verts, idx = [], []
    for k in range(n):
        verts.append(base + np.array([0, 0, np.float32(k * step)], np.float32))       # the rays come from z = -0.25 along +z
        idx.append([3 * k, 3 * k + 1, 3 * k + 2])
    mats = [dict(type=capi.MAT_LAMBERTIAN, rgb=(0.2, 0.5, 0.8), param=0.0, texture=None)]
    return dict(textures=[], materials=mats, meshes=[dict(vertices=np.concatenate(verts), indices=np.array(idx, np.uint32), material=0)], spheres=[])


def test_tie_window_follows_a_creeping_chain(hostsim, oracle, capi):
    """closest_so_far can creep past the fold's window through a chain of candidates ~1e-7 apart.  The fold notices
    (the last accepted t + 1e-7 reaches beyond the window) and folds again over a wider window: the winner is the
    reference's (the oracle walks its own LBVH without any window)."""
    desc = _creeping_chain_scene(capi)
    s = hostsim.build(desc)
    rs = oracle.RefScene(desc)
    assert np.array_equal(s.ranks(), rs.ranks())
    rng = np.random.default_rng(11)
    far_wins = 0
    for _ in range(200):
        o = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.5, 0.2), -0.25], np.float32)     # t ~ 0.25: ulp 3e-8, so 1e-7 steps resolve
        d = np.array([rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), 1.0], np.float32)
        ref_prim, ref_t = rs.trace(o, d)
        prim, t, took = hostsim.trace(s, o, d)
        assert prim == ref_prim and t == ref_t and took, (o, d, prim, ref_prim)
        far_wins += ref_prim >= 13                                                             # > 1.1e-6 behind the nearest triangle
    assert far_wins > 50


def test_division_by_a_sphere_radius_is_ieee_division(hostsim):
    """The hit normal (p - c) / radius (sphere.h:45) multiplies by a reciprocal that commit has proved exact for that radius
    (all 2^23 numerator significands): the quotients must be numpy's float32 divisions bit for bit, for ordinary numerators
    and for the ones the fast path hands to the plain division (zeros, denormals, tiny, huge, infinities), and radii outside
    [2^-20, 2^20] get no reciprocal at all."""
    import ctypes as C
    L = hostsim.L
    L.hostsim_div3_exact.argtypes = [C.c_float, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_float)]
    L.hostsim_div3_exact.restype = None
    rng = np.random.default_rng(7)
    n = 30000
    bits = rng.integers(0, 2**32, size=3 * n, dtype=np.uint64).astype(np.uint32)
    a = bits.view(np.float32).copy()
    a[np.isnan(a)] = np.float32(1.5)
    a[:12] = np.array([0.0, -0.0, 1e-45, -1e-40, 1e-30, 3e38, np.inf, -np.inf, 1.0, 2.0, 1000.0, 2.0 ** -60], dtype=np.float32)
    mid = (rng.standard_normal(3 * n) * 1000).astype(np.float32)        # what hit points on a sphere look like
    for radius, has_reciprocal in ((1000.0, True), (2.0, True), (0.2, True), (1.0, True), (3.7, True), (1.9999999, True), (1e-7, False), (3e6, False)):
        r32 = np.float32(radius)
        for numer in (a, mid):
            q = np.empty_like(numer)
            y = C.c_float()
            L.hostsim_div3_exact(float(r32), numer.ctypes.data, n, q.ctypes.data, C.byref(y))
            assert (y.value != 0.0) == has_reciprocal and (not has_reciprocal or np.float32(y.value) == np.float32(1.0) / r32)
            with np.errstate(all="ignore"):
                want = numer / r32
            assert np.array_equal(q.view(np.uint32), want.view(np.uint32)), radius
