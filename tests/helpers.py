"""Test-side helpers: the host single-stepper binding, tile (de)interleave in numpy, image metrics."""
import ctypes as C
import os

import numpy as np

from cudaraytracer_amd import capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


class HostSim:
    """tests/hostsim/libhostsim.so: the product's per-lane code compiled for the host (test-only)."""

    def __init__(self):
        self.L = capi.bind(C.CDLL(os.path.join(REPO, "tests", "hostsim", "libhostsim.so")))
        self.L.hostsim_render.argtypes = [C.c_void_p, C.POINTER(capi.Camera), C.POINTER(capi.RenderParams), C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.hostsim_trace.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.c_int,
                                         C.POINTER(C.c_int)]

    def build(self, desc):
        s = capi.Scene(self.L)
        for t in desc["textures"]:
            s.add_texture(t["type"], t["a"], t.get("b"), t.get("scale", 1.0))
        for m in desc["materials"]:
            s.add_material(m["type"], m["rgb"], m["param"], capi.NO_TEXTURE if m["texture"] is None else m["texture"])
        for me in desc["meshes"]:
            s.add_mesh(me["vertices"], me["indices"], me["material"])
        for sp in desc["spheres"]:
            s.add_sphere(sp["center"], sp["radius"], sp["material"])
        s.commit(capi.DEVICE_NONE)
        return s

    def render(self, scene, cam, width, height, spp, max_depth=10, seed=1984, sample_offset=0, ids_sample=0, tile_rank=0, tile_ranks=1, flags=0):
        p = capi.RenderParams(width=width, height=height, spp=spp, sample_offset=sample_offset, max_depth=max_depth, tile_rank=tile_rank,
                              tile_ranks=tile_ranks, flags=flags, ids_sample=ids_sample, sample_chunks=0, seed=seed, stream=None)
        out = np.zeros((height, width, 3), np.float32)
        ids = np.full((height, width), -2, np.int32)
        st = (C.c_uint64 * 8)()
        rc = self.L.hostsim_render(scene._h, C.byref(cam), C.byref(p), out.ctypes.data, ids.ctypes.data, st)
        assert rc == 0
        return out, ids, dict(rays=st[0], nodes=st[1], tri_tests=st[2], sphere_tests=st[3], tie_refolds=st[4], max_stack=st[5], shaded_hits=st[6])

    def trace(self, scene, o, d, tmin=0.001, force_tie_path=False):
        t, took = C.c_float(), C.c_int()
        prim = self.L.hostsim_trace(scene._h, (C.c_float * 3)(*o), (C.c_float * 3)(*d), tmin, C.byref(t), int(force_tie_path), C.byref(took))
        return prim, (t.value if prim >= 0 else None), bool(took.value)


def tile_index_map(width, height, tile_ranks):
    """For every (rank, slot, lane) of the compact layout: flat pixel index or -1 (crt_untile's mapping, in numpy)."""
    tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
    ntiles = tiles_x * tiles_y
    per = (ntiles + tile_ranks - 1) // tile_ranks
    m = np.full((tile_ranks, per, 64), -1, dtype=np.int64)
    lane = np.arange(64)
    for tile in range(ntiles):
        tx, ty = tile % tiles_x, tile // tiles_x
        i, j = tx * 8 + (lane & 7), ty * 8 + (lane >> 3)
        ok = (i < width) & (j < height)
        m[tile % tile_ranks, tile // tile_ranks, ok] = (j * width + i)[ok]
    return m


def to_compact(frame, rank, tile_ranks):
    """row-major [H,W,3] -> compact tile list of one rank (zeros outside the image)."""
    h, w, _ = frame.shape
    m = tile_index_map(w, h, tile_ranks)[rank]
    out = np.zeros((m.shape[0], 64, 3), dtype=frame.dtype)
    flat = frame.reshape(-1, 3)
    ok = m >= 0
    out[ok] = flat[m[ok]]
    return out.reshape(-1)


def untile_numpy(compact_all, width, height, tile_ranks):
    m = tile_index_map(width, height, tile_ranks)
    out = np.zeros((height * width, 3), dtype=compact_all.dtype)
    src = compact_all.reshape(tile_ranks, m.shape[1], 64, 3)
    ok = m >= 0
    out[m[ok]] = src[ok]
    return out.reshape(height, width, 3)


def psnr(a, b, peak=1.0):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)
