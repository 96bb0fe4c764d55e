"""ctypes binding of libcrt_hip.so (include/crt.h), used by tests/ and bench.py.

This is plumbing over the C ABI, not a second implementation: every call below lands in the HIP library.  If the
library is missing the import fails loudly -- there is no Python or CPU fallback for rendering.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRT_LIB: load another build of the same library (A/B runs of kernel variants during tuning)
LIB_PATH = os.environ.get("CRT_LIB") or os.path.join(_HERE, "libcrt_hip.so")

CRT_OK, CRT_ERR_INVALID, CRT_ERR_IO, CRT_ERR_HIP, CRT_ERR_LIMIT, CRT_ERR_NO_DEVICE = range(6)
TEX_SOLID, TEX_CHECKER = 0, 1
MAT_LAMBERTIAN, MAT_METAL, MAT_DIFFUSE_LIGHT, MAT_DIELECTRIC = 0, 1, 2, 3
BUILD_GPU_LBVH = 1
BUILD_GPU_SAH = 2
BUILD_AUTO = 4
OUT_ACCUM, OUT_COMPACT, COUNT_WORK, OUT_DEVICE = 1, 2, 4, 8
MULTI_TILE, MULTI_SAMPLE = 0, 1
NOALLOC_SLAB_BYTES = 32 << 20
# tests set this: re-read the CRT_* tuning knobs before every render (the library reads them at commit only)
RELOAD_TUNING_EACH_RENDER = False
DEVICE_NONE = -1
NO_TEXTURE = 0xFFFFFFFF


class CrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("crt error %d: %s" % (code, msg))
        self.code = code


class Camera(C.Structure):  # crt_camera
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("defocus_angle", C.c_float), ("focus_distance", C.c_float),
                ("center", C.c_float * 3), ("pixel00", C.c_float * 3), ("pixel_delta_u", C.c_float * 3), ("pixel_delta_v", C.c_float * 3),
                ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3), ("defocus_disk_u", C.c_float * 3),
                ("defocus_disk_v", C.c_float * 3), ("background", C.c_float * 3)]


class RenderParams(C.Structure):  # crt_render_params
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("sample_offset", C.c_uint32), ("max_depth", C.c_uint32),
                ("tile_rank", C.c_uint32), ("tile_ranks", C.c_uint32), ("flags", C.c_uint32), ("ids_sample", C.c_uint32), ("sample_chunks", C.c_uint32),
                ("seed", C.c_uint64), ("stream", C.c_void_p)]


class Stats(C.Structure):  # crt_stats
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("tri_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("tie_refolds", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("sched", C.c_uint64 * 10), ("kernel_ms", C.c_float), ("reserved", C.c_uint32)]


class ObjGroup(C.Structure):  # crt_obj_group
    _fields_ = [("object", C.c_char * 64), ("material", C.c_char * 64), ("Kd", C.c_float * 3), ("Ke", C.c_float * 3), ("Ns", C.c_float), ("Ni", C.c_float),
                ("d", C.c_float), ("illum", C.c_int32), ("first_triangle", C.c_uint32), ("num_triangles", C.c_uint32)]


class SceneInfo(C.Structure):  # crt_scene_info
    _fields_ = [("num_triangles", C.c_uint32), ("num_spheres", C.c_uint32), ("num_materials", C.c_uint32), ("num_textures", C.c_uint32),
                ("num_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("max_leaf_size", C.c_uint32), ("device", C.c_uint32),
                ("node_bytes", C.c_uint64), ("triangle_bytes", C.c_uint64), ("sphere_bytes", C.c_uint64),
                ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3), ("builder", C.c_uint32), ("build_ms", C.c_float),
                ("node_width", C.c_uint32), ("binary_depth", C.c_uint32), ("sah_cost", C.c_float)]


# every symbol include/crt.h declares (tests check that the library exports exactly these)
SYMBOLS = ["crt_last_error", "crt_version", "crt_load_obj", "crt_load_obj_ex", "crt_free", "crt_scene_create", "crt_scene_destroy", "crt_scene_add_texture",
           "crt_scene_add_material", "crt_scene_add_mesh", "crt_scene_add_sphere", "crt_scene_commit", "crt_scene_commit_ex", "crt_scene_commit_devices", "crt_scene_reload_tuning", "crt_scene_get_info",
           "crt_scene_get_ranks", "crt_camera_init", "crt_render", "crt_render_noalloc", "crt_render_multi", "crt_compact_floats", "crt_untile", "crt_finalize", "crt_quantize_rgb8", "crt_write_png"]

_lib = None


def lib():
    """Load libcrt_hip.so once.  Raises if it has not been built (python __graft_entry__.py / make -C cudaraytracer_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcrt_hip.so not built: run `make -C cudaraytracer_amd/csrc` (there is no CPU fallback)")
    _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def bind(L):
    """Declare the include/crt.h prototypes on a loaded library (libcrt_hip.so; tests/hostsim binds its own build too)."""
    f3 = C.POINTER(C.c_float)
    L.crt_last_error.restype = C.c_char_p
    L.crt_version.restype = C.c_int
    L.crt_load_obj.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)]
    L.crt_load_obj_ex.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32),
                                  C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(ObjGroup)), C.POINTER(C.c_uint32)]
    L.crt_free.argtypes = [C.c_void_p]
    L.crt_free.restype = None
    L.crt_scene_create.restype = C.c_void_p
    L.crt_scene_destroy.argtypes = [C.c_void_p]
    L.crt_scene_destroy.restype = None
    L.crt_scene_add_texture.argtypes = [C.c_void_p, C.c_int, f3, f3, C.c_float, C.POINTER(C.c_uint32)]
    L.crt_scene_add_material.argtypes = [C.c_void_p, C.c_int, f3, C.c_float, C.c_uint32, C.POINTER(C.c_uint32)]
    L.crt_scene_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    L.crt_scene_add_sphere.argtypes = [C.c_void_p, f3, C.c_float, C.c_uint32]
    L.crt_scene_commit.argtypes = [C.c_void_p, C.c_int]
    L.crt_scene_commit_ex.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    L.crt_scene_commit_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_uint32, C.c_uint32]
    L.crt_scene_reload_tuning.argtypes = [C.c_void_p]
    L.crt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
    L.crt_scene_get_ranks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.crt_camera_init.argtypes = [C.POINTER(Camera), C.c_uint32, f3, f3, f3, C.c_float, C.c_float, C.c_float, C.c_float]
    L.crt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.crt_render_noalloc.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p]
    L.crt_render_multi.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.crt_compact_floats.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.crt_compact_floats.restype = C.c_uint64
    L.crt_untile.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
    L.crt_finalize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
    L.crt_quantize_rgb8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
    L.crt_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
    return L


def _check(rc, L=None):
    if rc != CRT_OK:
        raise CrtError(rc, (L or lib()).crt_last_error().decode("utf-8", "replace"))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def load_obj(path):
    """crt_load_obj -> (vertices [n,6] float32 (position, normal), indices [m,3] uint32)."""
    L = lib()
    xyz = C.POINTER(C.c_float)()
    idx = C.POINTER(C.c_uint32)()
    nv, nt = C.c_uint32(), C.c_uint32()
    _check(L.crt_load_obj(os.fsencode(path), C.byref(xyz), C.byref(nv), C.byref(idx), C.byref(nt)))
    try:
        verts = np.ctypeslib.as_array(xyz, shape=(nv.value, 6)).copy()
        indices = np.ctypeslib.as_array(idx, shape=(nt.value, 3)).copy()
    finally:
        L.crt_free(xyz)
        L.crt_free(idx)
    return verts, indices


def camera(width, lookfrom, lookat, vup, vfov, aspect, defocus_angle, focus_dist):
    cam = Camera()
    _check(lib().crt_camera_init(C.byref(cam), width, _f3(lookfrom), _f3(lookat), _f3(vup), vfov, aspect, defocus_angle, focus_dist))
    return cam


class Scene:
    """Thin owner of a crt_scene handle."""

    def __init__(self, L=None):
        self._L = L or lib()
        self._h = self._L.crt_scene_create()
        if not self._h:
            raise MemoryError("crt_scene_create failed")

    def close(self):
        if self._h:
            self._L.crt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_texture(self, type_, a, b=None, scale=1.0):
        out = C.c_uint32()
        _check(self._L.crt_scene_add_texture(self._h, type_, _f3(a), _f3(b) if b is not None else None, scale, C.byref(out)), self._L)
        return out.value

    def add_material(self, type_, rgb, param=0.0, texture=NO_TEXTURE):
        out = C.c_uint32()
        _check(self._L.crt_scene_add_material(self._h, type_, _f3(rgb), param, texture, C.byref(out)), self._L)
        return out.value

    def add_mesh(self, vertices, indices, material):
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        _check(self._L.crt_scene_add_mesh(self._h, v.ctypes.data, v.shape[1], v.shape[0], i.ctypes.data, i.shape[0], material), self._L)

    def add_sphere(self, center, radius, material):
        _check(self._L.crt_scene_add_sphere(self._h, _f3(center), radius, material), self._L)

    def commit(self, device=0, flags=None):
        if flags is None:
            _check(self._L.crt_scene_commit(self._h, device), self._L)
        else:
            _check(self._L.crt_scene_commit_ex(self._h, device, flags), self._L)

    def commit_devices(self, devices, flags=0):
        """crt_scene_commit_devices: one copy of the scene per logical rank (the same device may be listed more than once)."""
        arr = (C.c_int * len(devices))(*devices)
        _check(self._L.crt_scene_commit_devices(self._h, arr, len(devices), flags), self._L)

    def reload_tuning(self):
        _check(self._L.crt_scene_reload_tuning(self._h), self._L)

    def info(self):
        info = SceneInfo()
        _check(self._L.crt_scene_get_info(self._h, C.byref(info)), self._L)
        return info

    def ranks(self):
        info = self.info()
        n = info.num_triangles + info.num_spheres
        out = np.empty(n, dtype=np.uint32)
        _check(self._L.crt_scene_get_ranks(self._h, out.ctypes.data, n), self._L)
        return out

    def render(self, cam, width, height, spp, max_depth=10, seed=1984, sample_offset=0, ids_sample=0, tile_rank=0, tile_ranks=1,
               flags=0, out_rgb=None, out_ids=None, want_ids=False, want_stats=True, stream=None, sample_chunks=0, multi=None):
        """crt_render (multi = MULTI_TILE / MULTI_SAMPLE: crt_render_multi over the devices of commit_devices).
        out_rgb / out_ids: numpy arrays (host) or integer device addresses; None allocates numpy outputs."""
        p = RenderParams(width=width, height=height, spp=spp, sample_offset=sample_offset, max_depth=max_depth, tile_rank=tile_rank,
                         tile_ranks=tile_ranks, flags=flags, ids_sample=ids_sample, sample_chunks=sample_chunks, seed=seed, stream=stream)
        L = self._L
        rgb_np = ids_np = None
        if out_rgb is None:
            n = int(L.crt_compact_floats(width, height, tile_ranks)) if (flags & OUT_COMPACT) else width * height * 3
            rgb_np = np.zeros(n, dtype=np.float32)
            out_rgb = rgb_np
        if out_ids is None and want_ids:
            ids_np = np.full(width * height, -2, dtype=np.int32)
            out_ids = ids_np
        rgb_ptr = out_rgb.ctypes.data if isinstance(out_rgb, np.ndarray) else out_rgb
        ids_ptr = (out_ids.ctypes.data if isinstance(out_ids, np.ndarray) else out_ids) if out_ids is not None else None
        st = Stats() if want_stats else None
        if RELOAD_TUNING_EACH_RENDER:
            self.reload_tuning()
        if multi is None:
            _check(L.crt_render(self._h, C.byref(cam), C.byref(p), rgb_ptr, ids_ptr, C.byref(st) if st is not None else None), L)
        else:
            _check(L.crt_render_multi(self._h, C.byref(cam), C.byref(p), multi, rgb_ptr, ids_ptr, C.byref(st) if st is not None else None), L)
        if rgb_np is not None and not (flags & OUT_COMPACT):
            rgb_np = rgb_np.reshape(height, width, 3)
        if ids_np is not None:
            ids_np = ids_np.reshape(height, width)
        return rgb_np if rgb_np is not None else out_rgb, ids_np if ids_np is not None else out_ids, st


def render_noalloc(scene, cam, width, height, spp, out_rgb, out_ids=None, out_rays=None, max_depth=10, seed=1984, sample_offset=0, ids_sample=0,
                   tile_rank=0, tile_ranks=1, flags=0, stream=None, sample_chunks=0):
    """crt_render_noalloc: device addresses only, asynchronous on `stream`, allocates nothing."""
    p = RenderParams(width=width, height=height, spp=spp, sample_offset=sample_offset, max_depth=max_depth, tile_rank=tile_rank,
                     tile_ranks=tile_ranks, flags=flags, ids_sample=ids_sample, sample_chunks=sample_chunks, seed=seed, stream=stream)
    if RELOAD_TUNING_EACH_RENDER:
        scene.reload_tuning()
    _check(scene._L.crt_render_noalloc(scene._h, C.byref(cam), C.byref(p), out_rgb, out_ids, out_rays), scene._L)


def untile(compact_all_ptr, width, height, tile_ranks, out_ptr, device=0, stream=None):
    _check(lib().crt_untile(compact_all_ptr, width, height, tile_ranks, out_ptr, device, stream))


def finalize(accum_ptr, width, height, total_spp, out_ptr, device=0, stream=None):
    _check(lib().crt_finalize(accum_ptr, width, height, total_spp, out_ptr, device, stream))


def compact_floats(width, height, tile_ranks):
    return int(lib().crt_compact_floats(width, height, tile_ranks))


def quantize_rgb8(rgb, width, height, device=0, stream=None):
    """crt_quantize_rgb8 (HIP kernel).  rgb: float32 numpy array (host, staged through the device) -> uint8 [H, W, 3]."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty((height, width, 3), dtype=np.uint8)
    _check(lib().crt_quantize_rgb8(rgb.ctypes.data, width, height, out.ctypes.data, device, stream))
    return out


def write_png(path, rgb8):
    """crt_write_png: uint8 [H, W, 3] -> PNG file (host code, no GPU needed)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w = rgb8.shape[:2]
    _check(lib().crt_write_png(os.fsencode(path), rgb8.ctypes.data, w, h))


def load_obj_groups(path):
    """crt_load_obj_ex -> (vertices [n,6] float32, indices [m,3] uint32, tri_group [m] uint32, groups: list of dicts)."""
    L = lib()
    v, i, g, gr = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(ObjGroup)()
    nv, nt, ng = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _check(L.crt_load_obj_ex(os.fsencode(path), C.byref(v), C.byref(nv), C.byref(i), C.byref(nt), C.byref(g), C.byref(gr), C.byref(ng)), L)
    try:
        verts = np.ctypeslib.as_array(v, shape=(nv.value, 6)).copy()
        idx = np.ctypeslib.as_array(i, shape=(nt.value, 3)).copy()
        tg = np.ctypeslib.as_array(g, shape=(nt.value,)).copy()
        groups = [dict(object=gr[k].object.decode(), material=gr[k].material.decode(), Kd=tuple(gr[k].Kd), Ke=tuple(gr[k].Ke), Ns=gr[k].Ns, Ni=gr[k].Ni,
                       d=gr[k].d, illum=gr[k].illum, first_triangle=gr[k].first_triangle, num_triangles=gr[k].num_triangles) for k in range(ng.value)]
    finally:
        for ptr in (v, i, g, gr):
            L.crt_free(C.cast(ptr, C.c_void_p))
    return verts, idx, tg, groups
