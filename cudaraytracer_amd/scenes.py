"""Scene descriptions for the BASELINE.json configurations, as plain data.

A description is a dict of lists (textures, materials, meshes, spheres) in insertion order; `build()` feeds it to the
HIP library through the C ABI.  tests/ and bench.py feed the very same description to the oracle, so both sides see
identical inputs.  Reference scene: Scene::launch_triangles_scene + kernel_triangles_scene (scene.h:158-177, 237-308):
every mesh triangle Lambertian(0.7, 0.3, 0.16), ground Sphere((0,-1000,-1), 1000, Lambertian(0.5)), light
Sphere((3,6.5,3), 2, DiffuseLight(15,15,15)); camera from init_camera (kernel.cu:30-42).
"""
import os

import numpy as np

from . import capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = os.path.join(REPO, "models")

# init_camera (kernel.cu:30-42): vfov 25, from (8,5,-6), at (-4,0,4), up (0,1,0), defocus 0.6, focus = |from|
REF_CAMERA = dict(lookfrom=(8.0, 5.0, -6.0), lookat=(-4.0, 0.0, 4.0), vup=(0.0, 1.0, 0.0), vfov=25.0, aspect=16.0 / 9.0, defocus_angle=0.6)
REF_SEED = 1984  # renderer.h:45


def focus_distance(lookfrom):
    x, y, z = (np.float32(c) for c in lookfrom)
    # glm::length(look_from - 0) with the library's dot order (x*x + y*y) + z*z; exact for the reference's (8,5,-6)
    return float(np.sqrt(np.float32(np.float32(np.float32(x * x) + np.float32(y * y)) + np.float32(z * z))))


def make_camera(width, **overrides):
    p = dict(REF_CAMERA)
    p.update(overrides)
    p.setdefault("focus_dist", focus_distance(p["lookfrom"]))
    cam = capi.camera(width, p["lookfrom"], p["lookat"], p["vup"], p["vfov"], p["aspect"], p["defocus_angle"], p["focus_dist"])
    if p.get("background") is not None:  # Camera::background is a public member (camera.h:23), default (0,0,0)
        for k in range(3):
            cam.background[k] = float(p["background"][k])
    # Camera::defocus_angle is a public member too (camera.h:20); the constructor's parameter of the same name shadows it, so
    # it stays 0.6 unless the caller writes it afterwards -- and it is the member that get_ray tests (camera.h:94)
    if p.get("member_defocus_angle") is not None:
        cam.defocus_angle = float(p["member_defocus_angle"])
    return p, cam


def _base():
    return dict(textures=[], materials=[], meshes=[], spheres=[])


def _add_ref_spheres(d):
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.5, 0.5, 0.5), param=0.0, texture=None))
    ground = len(d["materials"]) - 1
    d["materials"].append(dict(type=capi.MAT_DIFFUSE_LIGHT, rgb=(15.0, 15.0, 15.0), param=0.0, texture=None))
    light = len(d["materials"]) - 1
    d["spheres"].append(dict(center=(0.0, -1000.0, -1.0), radius=1000.0, material=ground))  # scene.h:164
    d["spheres"].append(dict(center=(3.0, 6.5, 3.0), radius=2.0, material=light))           # scene.h:165


def cat(path=None):
    """Config 1: models/cat.obj, Lambertian(0.7,0.3,0.16) + ground + emissive sphere (the reference scene as written)."""
    d = _base()
    v, i = capi.load_obj(path or os.path.join(MODELS, "cat.obj"))
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.7, 0.3, 0.16), param=0.0, texture=None))  # scene.h:148
    d["meshes"].append(dict(vertices=v[:, :3].copy(), indices=i, material=0))
    _add_ref_spheres(d)
    return d


def cat_checker(path=None):
    """Config 2: config 1 with the cat Lambertian(CheckerTexture(0.32, (.2,.3,.1), (.9,.9,.9))) (texture.h:23-44)."""
    d = _base()
    v, i = capi.load_obj(path or os.path.join(MODELS, "cat.obj"))
    d["textures"].append(dict(type=capi.TEX_CHECKER, a=(0.2, 0.3, 0.1), b=(0.9, 0.9, 0.9), scale=0.32))
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.0, 0.0, 0.0), param=0.0, texture=0))
    d["meshes"].append(dict(vertices=v[:, :3].copy(), indices=i, material=0))
    _add_ref_spheres(d)
    return d


def cats(path=None):
    """Config 3/5: models/cats.obj (three cats, 732 faces each, in file order): Metal(0.8, fuzz 0.1) on the first,
    Lambertian(0.7,0.3,0.16) on the other two, + ground + emissive sphere."""
    d = _base()
    v, i = capi.load_obj(path or os.path.join(MODELS, "cats.obj"))
    d["materials"].append(dict(type=capi.MAT_METAL, rgb=(0.8, 0.8, 0.8), param=0.1, texture=None))
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.7, 0.3, 0.16), param=0.0, texture=None))
    per = i.shape[0] // 3
    pos = v[:, :3].copy()
    d["meshes"].append(dict(vertices=pos, indices=i[:per], material=0))
    d["meshes"].append(dict(vertices=pos, indices=i[per:2 * per], material=1))
    d["meshes"].append(dict(vertices=pos, indices=i[2 * per:], material=1))
    _add_ref_spheres(d)
    return d


def cat_grid(instances=1366, side=37, pitch=3.5, path=None):
    """Config 4: `instances` translated copies of cat.obj on a side x side grid in the xz plane (row-major, centred on
    the origin), baked into one triangle soup: 1366 x 732 = 999,912 triangles.  Deterministic, no RNG."""
    d = _base()
    v, i = capi.load_obj(path or os.path.join(MODELS, "cat.obj"))
    pos = v[:, :3].astype(np.float32)
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.7, 0.3, 0.16), param=0.0, texture=None))
    k = np.arange(instances)
    off = np.zeros((instances, 3), dtype=np.float32)
    off[:, 0] = ((k % side).astype(np.float32) - np.float32((side - 1) / 2)) * np.float32(pitch)
    off[:, 2] = ((k // side).astype(np.float32) - np.float32((side - 1) / 2)) * np.float32(pitch)
    verts = (pos[None, :, :] + off[:, None, :]).reshape(-1, 3).astype(np.float32)
    idx = (i[None, :, :].astype(np.uint32) + (k[:, None, None] * pos.shape[0]).astype(np.uint32)).reshape(-1, 3)
    d["meshes"].append(dict(vertices=verts, indices=idx, material=0))
    _add_ref_spheres(d)
    return d


def terrain(n=50):
    """A regular height field of 2 n^2 triangles over [-4, 4]^2 (Lambertian) + the reference spheres: SAH gives it a balanced,
    shallow tree (n = 50: 5000 triangles, depth 12; n = 64: 8192 triangles, depth 14) -- a tree of at most 14 levels whose
    leaf references do NOT fit 16 bits, the case the 16-bit-stack kernels must leave to the 32-bit ones."""
    d = _base()
    xs = np.linspace(-4.0, 4.0, n + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = (0.4 * np.sin(1.7 * X) * np.cos(1.3 * Z) + 0.5).astype(np.float32)
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    a = (np.arange(n)[:, None] * (n + 1) + np.arange(n)[None, :]).reshape(-1).astype(np.uint32)
    idx = np.concatenate([np.stack([a, a + 1, a + n + 1], -1), np.stack([a + 1, a + n + 2, a + n + 1], -1)], 0).astype(np.uint32)
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.3, 0.6, 0.2), param=0.0, texture=None))
    d["meshes"].append(dict(vertices=verts, indices=idx, material=0))
    _add_ref_spheres(d)
    return d


# camera for config 4: the reference direction of view, pulled back along it to frame the 130-unit grid
GRID_CAMERA = dict(lookfrom=(80.0, 50.0, -60.0), lookat=(-4.0, 0.0, 4.0))


def random_spheres(seed=1984):
    """Row N3: the reference's sphere scene (kernel_create_world, scene.h:53-90): ground sphere, a 22 x 22 field of r = 0.2
    spheres (80 % Lambertian(RND*RND), 15 % Metal(0.5(1+RND), fuzz 0.5 RND), 5 % Dielectric(1.5)) and three r = 1
    spheres, 488 in all, no triangles.  The reference draws RND = curand_uniform from one XORWOW state; here the draws
    come, in the same order, from numpy's Philox generator keyed on `seed` (same distribution, (0,1], other numbers)."""
    rng = np.random.Generator(np.random.Philox(key=seed))

    def rnd():
        return float(np.float32(1.0) - np.float32(rng.random(dtype=np.float32)))  # (0, 1]

    d = _base()

    def sphere(center, radius, mtype, rgb, param):
        d["materials"].append(dict(type=mtype, rgb=tuple(float(np.float32(c)) for c in rgb), param=float(np.float32(param)), texture=None))
        d["spheres"].append(dict(center=tuple(float(np.float32(c)) for c in center), radius=float(np.float32(radius)), material=len(d["materials"]) - 1))

    sphere((0.0, -1000.0, -1.0), 1000.0, capi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = rnd()
            center = (a + rnd(), 0.2, b + rnd())
            if choose < 0.8:
                sphere(center, 0.2, capi.MAT_LAMBERTIAN, (rnd() * rnd(), rnd() * rnd(), rnd() * rnd()), 0.0)
            elif choose < 0.95:
                sphere(center, 0.2, capi.MAT_METAL, (0.5 * (1.0 + rnd()), 0.5 * (1.0 + rnd()), 0.5 * (1.0 + rnd())), 0.5 * rnd())
            else:
                sphere(center, 0.2, capi.MAT_DIELECTRIC, (1.0, 1.0, 1.0), 1.5)
    sphere((0.0, 1.0, 0.0), 1.0, capi.MAT_DIELECTRIC, (1.0, 1.0, 1.0), 1.5)
    sphere((-4.0, 1.0, 0.0), 1.0, capi.MAT_LAMBERTIAN, (0.4, 0.2, 0.1), 0.0)
    sphere((4.0, 1.0, 0.0), 1.0, capi.MAT_METAL, (0.7, 0.6, 0.5), 0.0)
    return d


# camera for the sphere scene: the book's view of it; the reference's Camera::background stays a member one can set, and
# with its default (0,0,0) and no light in this scene the frame would be black, so a sky colour is used
SPHERES_CAMERA = dict(lookfrom=(13.0, 2.0, 3.0), lookat=(0.0, 0.0, 0.0), vfov=20.0, background=(0.7, 0.8, 1.0))


def build(desc, device=0, flags=None):
    """Description -> committed capi.Scene (flags: capi.BUILD_GPU_LBVH for the device-side builder)."""
    s = capi.Scene()
    for t in desc["textures"]:
        s.add_texture(t["type"], t["a"], t.get("b"), t.get("scale", 1.0))
    # materials without an explicit texture get their own SolidColor, appended after the explicit ones, like the oracle
    for m in desc["materials"]:
        s.add_material(m["type"], m["rgb"], m["param"], capi.NO_TEXTURE if m["texture"] is None else m["texture"])
    for me in desc["meshes"]:
        s.add_mesh(me["vertices"], me["indices"], me["material"])
    for sp in desc["spheres"]:
        s.add_sphere(sp["center"], sp["radius"], sp["material"])
    s.commit(device, flags)
    return s


def num_primitives(desc):
    return sum(m["indices"].shape[0] for m in desc["meshes"]) + len(desc["spheres"])
