"""A parity campaign, longer than the test suite's sweep: random scenes / frame sizes / sample counts / depths / seeds / sample windows /
partitions / kernel variants (CRT_NODE_CACHE, CRT_CHUNK_MODE) / builders, every frame compared with the oracle -- primary-hit ids and
ray count exactly, the image exactly (whole pixels, or one sample per unit) or against the oracle's chunk sums added in chunk order.
usage: python tools/fuzz_parity.py [seconds] [rng seed]     (prints one line per mismatch, a summary at the end; exit 1 on mismatch)"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from cudaraytracer_amd import capi, scenes
from oracle import binding as oracle

capi.RELOAD_TUNING_EACH_RENDER = True
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
NAMES = ["cat", "cats", "cat_checker", "random_spheres", "terrain", "spheres_on_a_floor"]
BUILDERS = [0, capi.BUILD_GPU_SAH, capi.BUILD_GPU_LBVH]
built, refs = {}, {}


def spheres_on_a_floor():
    """Spheres in the tree among triangles: 40 spheres hovering over, resting on, or sunk into a two-triangle floor (heights from -r
    to a few 1e-4 above it), some touching each other, a glass one and a metal one among them, a light above; seen from above and
    from BELOW the floor (a sphere's computed root can lie before its own box)."""
    r = np.random.default_rng(99)
    d = scenes._base()
    d["materials"] += [dict(type=capi.MAT_LAMBERTIAN, rgb=(0.6, 0.5, 0.4), param=0.0, texture=None), dict(type=capi.MAT_DIELECTRIC, rgb=(1, 1, 1), param=1.5, texture=None),
                       dict(type=capi.MAT_METAL, rgb=(0.8, 0.8, 0.7), param=0.1, texture=None), dict(type=capi.MAT_DIFFUSE_LIGHT, rgb=(4, 4, 4), param=0.0, texture=None)]
    d["spheres"] = []
    for k in range(40):
        rad = float(r.choice([0.2, 0.2, 0.35, 0.05]))
        lift = float(r.choice([0.0, 2e-5, 2e-4, 1e-3, -0.5 * rad, -rad]))
        d["spheres"].append(dict(center=(float(r.uniform(-3, 3)), rad + lift, float(r.uniform(-2, 2))), radius=rad, material=int(r.choice([0, 0, 1, 2]))))
    d["spheres"].append(dict(center=(0.0, 6.0, 0.0), radius=1.5, material=3))
    v = np.array([[-30, 0, -30], [30, 0, -30], [30, 0, 30], [-30, 0, 30]], np.float32)
    d["meshes"].append(dict(vertices=v, indices=np.array([[0, 1, 2], [0, 2, 3]], np.uint32), material=0))
    return d


def scene_for(name, flags):
    if (name, flags) not in built:
        desc = spheres_on_a_floor() if name == "spheres_on_a_floor" else getattr(scenes, name)() if name != "terrain" else scenes.terrain(24)
        built[(name, flags)] = (desc, scenes.build(desc, 0, flags if name not in ("random_spheres", "spheres_on_a_floor") else 0))
    return built[(name, flags)]


def ref_for(name, desc):
    if name not in refs:
        refs[name] = oracle.RefScene(desc)
    return refs[name]


t0, trials, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    name = NAMES[int(rng.integers(len(NAMES)))]
    flags = BUILDERS[int(rng.integers(len(BUILDERS)))]
    desc, s = scene_for(name, flags)
    rs = ref_for(name, desc)
    width = int(rng.integers(9, 260))
    height = int(rng.integers(1, max(2, int(width / (16 / 9)) + 1)))
    spp = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12]))
    divisors = [d for d in range(1, spp + 1) if spp % d == 0]
    chunks = int(rng.choice([0, spp] + divisors))
    kw = dict(max_depth=int(rng.integers(1, 12)), seed=int(rng.integers(0, 2 ** 40)), sample_offset=int(rng.integers(0, 50)))
    kw["ids_sample"] = kw["sample_offset"] + int(rng.integers(0, spp))
    camera = dict(scenes.SPHERES_CAMERA) if name == "random_spheres" else {}
    if name == "spheres_on_a_floor":
        dist = float(10.0 ** rng.uniform(0.5, 2.0))
        az, el = float(rng.uniform(0, 2 * np.pi)), float(rng.uniform(0.02, 1.5)) * float(rng.choice([-1.0, 1.0]))
        camera = dict(lookfrom=(dist * np.cos(az) * np.cos(el), dist * np.sin(el), dist * np.sin(az) * np.cos(el)), lookat=(float(rng.uniform(-2, 2)), 0.0, float(rng.uniform(-1, 1))),
                      vfov=float(np.degrees(2 * np.arctan(4.0 / dist))), background=(0.6, 0.7, 0.9))
    if rng.random() < 0.3:
        camera["defocus_angle"] = float(rng.choice([0.0, 2.0, 5.0]))
    if rng.random() < 0.15:
        camera["member_defocus_angle"] = 0.0
    os.environ["CRT_NODE_CACHE"] = str(rng.choice(["0", "1", "2", "3", "4"]))
    os.environ["CRT_CHUNK_MODE"] = str(rng.choice(["0", "1", "2"]))
    params, cam = scenes.make_camera(width, **camera)
    ocam = oracle.make_camera(params, width)
    ranks = int(rng.choice([1, 1, 2, 3]))
    trials += 1
    try:
        if ranks == 1:
            rgb, ids, st = s.render(cam, width, height, spp, want_ids=True, sample_chunks=chunks, **kw)
        else:  # a tile split: compact lists of every rank, untiled on the host
            from tests import helpers
            parts, rays = [], 0
            ids = None
            for r in range(ranks):
                part, pid, pst = s.render(cam, width, height, spp, want_ids=True, sample_chunks=chunks, tile_rank=r, tile_ranks=ranks, flags=capi.OUT_COMPACT, **kw)
                parts.append(part); rays += pst.rays
                ids = pid if ids is None else np.where(pid != ids, np.maximum(pid, ids), ids) if False else ids
            rgb = helpers.untile_numpy(np.concatenate(parts), width, height, ranks)
            ids = None

            class _S: pass
            st = _S(); st.rays = rays
        if chunks in (0, spp) or spp == 1:
            fb, rids, _, rst = rs.render(ocam, width, height, spp, **kw)
            ok_img = np.array_equal(np.asarray(rgb).reshape(fb.shape), fb)
        else:  # chunk sums added in chunk order, then / spp and gamma
            total, nrays = None, 0
            for c in range(chunks):
                k2 = dict(kw); k2["sample_offset"] = kw["sample_offset"] + c * (spp // chunks)
                k2.pop("ids_sample")
                _, _, acc, cst = rs.render(ocam, width, height, spp // chunks, want_ids=False, want_accum=True, **k2)
                total = acc.copy() if total is None else (total + acc).astype(np.float32)
                nrays += cst.rays
            v = (total / np.float32(spp)).astype(np.float32)
            fb = np.where(v > 0, np.sqrt(v), np.float32(0)).astype(np.float32)
            _, rids, _, _ = rs.render(ocam, width, height, 1, sample_offset=kw["ids_sample"], ids_sample=kw["ids_sample"], max_depth=kw["max_depth"], seed=kw["seed"])

            class _R: pass
            rst = _R(); rst.rays = nrays
            ok_img = np.array_equal(np.asarray(rgb).reshape(fb.shape), fb)
        ok = ok_img and st.rays == rst.rays and (ids is None or np.array_equal(ids, rids))
    except Exception as e:  # an error is a finding too
        ok = False
        print("EXCEPTION", repr(e)[:200])
    if not ok:
        bad += 1
        print("MISMATCH", dict(scene=name, builder=flags, w=width, h=height, spp=spp, chunks=chunks, ranks=ranks, camera=camera,
                               node_cache=os.environ["CRT_NODE_CACHE"], chunk_mode=os.environ["CRT_CHUNK_MODE"], **kw), flush=True)
    if trials % 5000 == 0:
        print("... %d trials, %d mismatches, %.0f s" % (trials, bad, time.time() - t0), flush=True)
print("fuzz: %d trials, %d mismatches, %.0f s" % (trials, bad, time.time() - t0))
sys.exit(1 if bad else 0)
