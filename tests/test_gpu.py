"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the oracle.

Bars: primary-hit primitive ids bit-exact; images bit-exact too (both sides execute the same IEEE operation sequence
and the same Philox streams) -- the stated tolerance for the image is therefore PSNR = inf, checked as array equality,
with the PSNR printed if it ever fails; path-segment (ray) counts equal.
"""
import hashlib
import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_scenes(capi, scenes):
    cache = {}

    def get(name):
        if name not in cache:
            desc = getattr(scenes, name)()
            cache[name] = (desc, scenes.build(desc, 0))
        return cache[name]
    return get


def _compare(oracle, scenes, gpu_scenes, name, width, height, spp, camera=None, **kw):
    desc, s = gpu_scenes(name)
    params, cam = scenes.make_camera(width, **(camera or {}))
    rgb, ids, st = s.render(cam, width, height, spp, want_ids=True, **kw)
    rs = oracle.RefScene(desc)
    okw = {k: v for k, v in kw.items() if k in ("max_depth", "seed", "sample_offset", "ids_sample")}
    fb, rids, _, rst = rs.render(oracle.make_camera(params, width), width, height, spp, **okw)
    assert np.array_equal(ids, rids), "ids differ in %d pixels" % int((ids != rids).sum())
    assert np.array_equal(rgb, fb), "image differs: PSNR %.1f dB, max abs %g" % (helpers.psnr(rgb, fb), float(np.abs(rgb - fb).max()))
    assert st.rays == rst.rays
    return rgb, ids, st, rst


def test_config1_cat_400x400_16spp(oracle, scenes, gpu_scenes):
    rgb, ids, st, rst = _compare(oracle, scenes, gpu_scenes, "cat", 400, 400, 16)
    g = np.load(os.path.join(helpers.GOLDEN, "cat_400x400_primary_ids.npz"))
    assert np.array_equal(ids, g["ids"].astype(np.int32)) and st.rays == int(g["rays_16spp"])
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == str(g["sha256_16spp"])


@pytest.mark.parametrize("name,width,height,spp,kw", [
    ("cats", 480, 270, 4, {}),
    ("cat_checker", 480, 270, 4, dict(max_depth=8)),
    ("cats", 203, 117, 3, dict(seed=7, sample_offset=11, ids_sample=12)),   # ragged size, other seed / sample window
    ("cat", 64, 36, 1, dict(max_depth=1)),
])
def test_small_renders_bit_exact(oracle, scenes, gpu_scenes, name, width, height, spp, kw):
    _compare(oracle, scenes, gpu_scenes, name, width, height, spp, **kw)


def test_cats_1080p_crop_against_golden_and_oracle(oracle, scenes, gpu_scenes):
    desc, s = gpu_scenes("cats")
    params, cam = scenes.make_camera(1920)
    rgb, ids, st = s.render(cam, 1920, 1080, 1, want_ids=True)
    g = np.load(os.path.join(helpers.GOLDEN, "cats_1080p_crop_ids.npz"))
    x0, y0, x1, y1 = (int(v) for v in g["window"])
    assert np.array_equal(ids[y0:y1, x0:x1], g["ids"].astype(np.int32))
    assert hashlib.sha256(np.ascontiguousarray(rgb[y0:y1, x0:x1]).tobytes()).hexdigest() == str(g["sha256"])
    # and the whole 1080p frame of primary ids against the oracle run here
    rs = oracle.RefScene(desc)
    fb, rids, _, rst = rs.render(oracle.make_camera(params, 1920), 1920, 1080, 1)
    assert np.array_equal(ids, rids) and np.array_equal(rgb, fb) and st.rays == rst.rays


def test_work_counters_and_tie_path(oracle, scenes, gpu_scenes, capi):
    desc, s = gpu_scenes("cats")
    _, cam = scenes.make_camera(480)
    rgb0, _, st0 = s.render(cam, 480, 270, 4)
    rgb1, _, st1 = s.render(cam, 480, 270, 4, flags=capi.COUNT_WORK)
    assert np.array_equal(rgb0, rgb1) and st0.rays == st1.rays and st0.nodes == 0
    assert st1.nodes > st1.rays and st1.tri_tests > 0 and st1.sphere_tests >= 2 * st1.rays
    assert st1.tie_refolds == st0.tie_refolds


def test_tie_scene_on_gpu(oracle, capi, scenes):
    from tests.test_host import _tie_scene
    desc = _tie_scene(capi)
    s = scenes.build(desc, 0)
    cam_params = dict(lookfrom=(0.0, 0.0, 4.0), lookat=(0.0, 0.0, 0.0), vup=(0.0, 1.0, 0.0), vfov=40.0, aspect=16 / 9, defocus_angle=0.6, focus_dist=4.0)
    cam = capi.camera(128, **cam_params)
    rgb, ids, st = s.render(cam, 128, 72, 8, want_ids=True)
    rs = oracle.RefScene(desc)
    fb, rids, _, rst = rs.render(oracle.make_camera(cam_params, 128), 128, 72, 8)
    assert np.array_equal(ids, rids) and np.array_equal(rgb, fb) and st.rays == rst.rays and st.tie_refolds > 0


def test_tile_partition_and_gather_are_bitwise_identical(scenes, gpu_scenes, capi):
    import torch
    _, s = gpu_scenes("cats")
    W, H, spp = 500, 281, 2                                    # ragged
    _, cam = scenes.make_camera(W)
    full, ids_full, st_full = s.render(cam, W, H, spp, want_ids=True)
    for ranks in (2, 3, 8):
        n = capi.compact_floats(W, H, ranks)
        allc = torch.zeros(ranks * n, dtype=torch.float32, device="cuda")
        rays = 0
        rowmajor = np.zeros((H, W, 3), np.float32)
        for r in range(ranks):
            _, _, st = s.render(cam, W, H, spp, tile_rank=r, tile_ranks=ranks, flags=capi.OUT_COMPACT, out_rgb=allc[r * n:].data_ptr())
            rays += st.rays
            part, _, _ = s.render(cam, W, H, spp, tile_rank=r, tile_ranks=ranks, out_rgb=rowmajor)
        out = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
        capi.untile(allc.data_ptr(), W, H, ranks, out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().reshape(H, W, 3), full)
        assert np.array_equal(rowmajor, full) and rays == st_full.rays
        assert np.array_equal(helpers.untile_numpy(allc.cpu().numpy(), W, H, ranks), full)


def test_sample_split_accumulate_and_finalize(scenes, gpu_scenes, capi):
    import torch
    _, s = gpu_scenes("cats")
    W, H = 320, 180
    _, cam = scenes.make_camera(W)
    full, _, st = s.render(cam, W, H, 8)
    # one call, accumulate + finalize == direct (same sum order)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    s.render(cam, W, H, 8, flags=capi.OUT_ACCUM, out_rgb=acc.data_ptr())
    out = torch.empty_like(acc)
    capi.finalize(acc.data_ptr(), W, H, 8, out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(H, W, 3), full)
    # split 4 ways by sample index: sum order differs -> tolerance: a few ulp of the sum (stated: rtol 1e-6 on the linear sum)
    parts = torch.zeros(4, H * W * 3, dtype=torch.float32, device="cuda")
    rays = 0
    for k in range(4):
        _, _, stk = s.render(cam, W, H, 2, sample_offset=2 * k, flags=capi.OUT_ACCUM, out_rgb=parts[k].data_ptr())
        rays += stk.rays
    assert rays == st.rays
    torch.cuda.synchronize()
    summed = parts.sum(0)
    assert torch.allclose(summed, acc, rtol=1e-6, atol=1e-7)
    capi.finalize(summed.data_ptr(), W, H, 8, out.data_ptr())
    torch.cuda.synchronize()
    assert helpers.psnr(out.cpu().numpy().reshape(H, W, 3), full) > 100.0


def test_device_and_host_outputs_agree(scenes, gpu_scenes):
    import torch
    _, s = gpu_scenes("cat")
    W, H = 256, 144
    _, cam = scenes.make_camera(W)
    host, ids_host, _ = s.render(cam, W, H, 2, want_ids=True)
    dev = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
    ids = torch.empty(H * W, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    s.render(cam, W, H, 2, out_rgb=dev.data_ptr(), out_ids=ids.data_ptr(), want_stats=False, stream=stream.cuda_stream)
    stream.synchronize()
    assert np.array_equal(dev.cpu().numpy().reshape(H, W, 3), host) and np.array_equal(ids.cpu().numpy().reshape(H, W), ids_host)


def test_full_size_properties_1080p_64spp(scenes, gpu_scenes, capi):
    """BASELINE headline size (cats.obj, 1920x1080, 64 spp): too slow for the oracle in a test, so size-independent
    properties: run-to-run bitwise determinism, tile-split == single launch bitwise, sample-split ray counts add up,
    and a 16-sample prefix equals the oracle-checked kernel's output for those samples."""
    import torch
    _, s = gpu_scenes("cats")
    W, H, spp = 1920, 1080, 64
    _, cam = scenes.make_camera(W)
    a = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    _, _, st_a = s.render(cam, W, H, spp, out_rgb=a.data_ptr())
    rays = 0
    for r in range(4):
        _, _, st = s.render(cam, W, H, spp, tile_rank=r, tile_ranks=4, out_rgb=b.data_ptr())
        rays += st.rays
    torch.cuda.synchronize()
    assert torch.equal(a, b) and rays == st_a.rays
    assert hashlib.sha256(a.cpu().numpy().tobytes()).hexdigest() == hashlib.sha256(b.cpu().numpy().tobytes()).hexdigest()
    r0 = s.render(cam, W, H, 32, sample_offset=0, flags=capi.OUT_ACCUM, out_rgb=b.data_ptr())[2].rays
    r1 = s.render(cam, W, H, 32, sample_offset=32, flags=capi.OUT_ACCUM, out_rgb=b.data_ptr())[2].rays
    assert r0 + r1 == st_a.rays
    img = a.cpu().numpy().reshape(H, W, 3)
    assert np.isfinite(img).all() and img.min() >= 0 and img[600:, :].mean() > 0.01


def test_grid_scene_1m_triangles_sample(oracle, scenes, capi, node_cache):
    """BASELINE config 4 geometry (999,912 triangles): BVH depth within the stack, and a window of primary rays + a
    low-resolution full path render bit-exact against the oracle (reference traversal, no culling -> small sizes)."""
    desc = scenes.cat_grid()
    s = scenes.build(desc, 0)
    info = s.info()
    assert info.num_triangles == 999912 and info.bvh_depth <= 32
    W, H = 192, 108
    params, cam = scenes.make_camera(W, **scenes.GRID_CAMERA)
    rgb, ids, st = s.render(cam, W, H, 1, want_ids=True)
    rs = oracle.RefScene(desc)
    fb, rids, _, rst = rs.render(oracle.make_camera(params, W), W, H, 1)
    assert np.array_equal(ids, rids) and np.array_equal(rgb, fb) and st.rays == rst.rays
    assert (ids >= 0).mean() > 0.5 and (ids < 999912).mean() > 0.2


@pytest.mark.parametrize("n", [45, 50, 64])
def test_shallow_trees_around_the_16_bit_reference_limit(oracle, scenes, capi, node_cache, n):
    """The 14-level kernels keep 16-bit stack entries (crt_kernels.hip: LdsStack16), which only a tree with at most 4096
    triangle slots and fewer than 32767 nodes may use.  Height fields of 4050 (fits), 5000 and 8192 triangles (do not fit;
    depth 12 and 14, so only the reference width sends them to the 32-bit kernels), every kernel variant, bit-exact."""
    desc = scenes.terrain(n)
    s = scenes.build(desc, 0)
    info = s.info()
    assert info.num_triangles == 2 * n * n and info.bvh_depth <= 14
    W, H, spp = 96, 54, 2
    params, cam = scenes.make_camera(W)
    rs = oracle.RefScene(desc)
    fb, rids, _, rst = rs.render(oracle.make_camera(params, W), W, H, spp)
    for chunks in (0, spp):
        rgb, ids, st = s.render(cam, W, H, spp, want_ids=True, sample_chunks=chunks)
        assert np.array_equal(ids, rids) and np.array_equal(rgb.reshape(fb.shape), fb) and st.rays == rst.rays, chunks
    assert ((ids >= 0) & (ids < info.num_triangles)).mean() > 0.05   # the camera does see the mesh


def test_frame_does_not_depend_on_the_schedule_knobs(monkeypatch, scenes, gpu_scenes):
    """A lane's own operation order is the sequential one whatever the wave does around it, so the knobs that only move
    work between lanes and waves -- units per trip to a block counter, the shade-phase threshold, waves per CU, interleaved
    instead of banded block counters -- must leave every pixel, id and the ray count untouched (whole pixels and one
    sample per unit, a ragged frame that ends in a partial block)."""
    _, s = gpu_scenes("cats")
    width, height, spp = 203, 117, 8
    _, cam = scenes.make_camera(width)
    for chunks in (1, spp):
        ref = s.render(cam, width, height, spp, want_ids=True, sample_chunks=chunks)
        for knob, values in (("CRT_FETCH_BLOCK", ("64", "256", "4096", "65536")), ("CRT_SCHED_THRESHOLD", ("1", "33", "64")),
                             ("CRT_WAVES_PER_CU", ("4", "12")), ("CRT_STRIPE_BANDS", ("0",))):
            for v in values:
                monkeypatch.setenv(knob, v)
                rgb, ids, st = s.render(cam, width, height, spp, want_ids=True, sample_chunks=chunks)
                monkeypatch.delenv(knob, raising=False)
                assert np.array_equal(rgb, ref[0]) and np.array_equal(ids, ref[1]) and st.rays == ref[2].rays, (chunks, knob, v)


def test_context_pool_kernel_matches_default_kernel(monkeypatch, scenes, gpu_scenes):
    """CRT_KERNEL=1 selects the experimental LDS context-pool schedule (crt_kernels.hip: render_pool_kernel): other
    lanes render other pixels at other times, yet every pixel, id and the ray count must be identical."""
    for name, width, height, spp, kw in [("cats", 256, 144, 8, {}), ("cat_checker", 203, 117, 5, dict(max_depth=6, ids_sample=3)),
                                         ("cats", 64, 40, 3, dict(tile_rank=1, tile_ranks=3))]:
        _, s = gpu_scenes(name)
        _, cam = scenes.make_camera(width)
        monkeypatch.delenv("CRT_KERNEL", raising=False)
        rgb0, ids0, st0 = s.render(cam, width, height, spp, want_ids=True, **kw)
        monkeypatch.setenv("CRT_KERNEL", "1")
        rgb1, ids1, st1 = s.render(cam, width, height, spp, want_ids=True, **kw)
        monkeypatch.delenv("CRT_KERNEL", raising=False)
        assert np.array_equal(rgb0, rgb1) and np.array_equal(ids0, ids1) and st0.rays == st1.rays


def test_sample_chunks_match_chunked_sum_of_oracle(oracle, scenes, gpu_scenes, node_cache, chunk_mode):
    """sample_chunks = n: the work unit is (pixel, chunk of spp/n samples); every sample is the same, each chunk is
    summed in sample order and the chunk sums are added in chunk order.  The oracle's raw sums per chunk, combined the
    same way in float32, must reproduce the image bit for bit; against the unchunked image the tolerance is float
    association only (rtol 3e-7 on the gamma-encoded value, as for the sample-split mode)."""
    name, width, height, spp, n = "cats", 160, 90, 12, 4
    desc, s = gpu_scenes(name)
    params, cam = scenes.make_camera(width)
    rgb1, ids1, st1 = s.render(cam, width, height, spp, want_ids=True, ids_sample=7)
    rgbn, idsn, stn = s.render(cam, width, height, spp, want_ids=True, ids_sample=7, sample_chunks=n)
    assert stn.rays == st1.rays and np.array_equal(idsn, ids1)
    assert np.allclose(rgbn, rgb1, rtol=3e-7, atol=1e-7)
    rs = oracle.RefScene(desc)
    ocam = oracle.make_camera(params, width)
    total = None
    for c in range(n):
        _, _, acc, _ = rs.render(ocam, width, height, spp // n, sample_offset=c * (spp // n), want_ids=False, want_accum=True)
        total = acc.copy() if total is None else (total + acc).astype(np.float32)
    v = (total / np.float32(spp)).astype(np.float32)
    expect = np.where(v > 0, np.sqrt(v, dtype=np.float32), np.float32(0)).astype(np.float32)
    assert np.array_equal(rgbn.reshape(expect.shape), expect)
    # raw sums + compact tile lists go through the same slabs
    from cudaraytracer_amd import capi
    accn, _, _ = s.render(cam, width, height, spp, flags=capi.OUT_ACCUM, sample_chunks=n)
    assert np.array_equal(accn.reshape(total.shape), total)
    parts = [s.render(cam, width, height, spp, tile_rank=r, tile_ranks=3, flags=capi.OUT_COMPACT, sample_chunks=n)[0] for r in range(3)]
    full = helpers.untile_numpy(np.concatenate(parts), width, height, 3)
    assert np.array_equal(full.reshape(expect.shape), expect)


@pytest.mark.parametrize("name,width,height,spp", [("cats", 97, 45, 8), ("cat_checker", 64, 36, 4), ("cats", 43, 19, 64),
                                                   ("cats", 97, 45, 3), ("cat", 120, 67, 10), ("cats", 75, 41, 12), ("cat", 33, 17, 100)])
def test_one_sample_per_chunk_is_the_sequential_sum(oracle, scenes, gpu_scenes, chunk_mode, name, width, height, spp):
    """sample_chunks = spp: every work unit is one sample, and a pixel's single samples are added in sample order starting
    from 0 -- the float operations of the reference's loop (renderer.h:26-32) -- by the chunk-sum pass (slabs) or by the wave
    that rendered them (rings).  So the frame must be the oracle's image (the sequential sum) bit for bit, ids and ray count
    included: the parallel decomposition bench.py times costs no exactness.  Any sample count (3, 10 -- the reference's own
    literal, kernel.cu:49 --, 12, 100: no power of two is needed), ragged frame sizes, raw sums and compact tile lists as well."""
    from cudaraytracer_amd import capi
    desc, s = gpu_scenes(name)
    params, cam = scenes.make_camera(width)
    fb, ids, acc, rst = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp, want_accum=True)
    rgb, gids, st = s.render(cam, width, height, spp, want_ids=True, sample_chunks=spp)
    assert st.rays == rst.rays and np.array_equal(gids, ids)
    assert np.array_equal(rgb.reshape(fb.shape), fb), helpers.psnr(rgb.reshape(fb.shape), fb)
    raw = s.render(cam, width, height, spp, flags=capi.OUT_ACCUM, sample_chunks=spp)[0]
    assert np.array_equal(raw.reshape(acc.shape), acc)
    parts = [s.render(cam, width, height, spp, tile_rank=r, tile_ranks=3, flags=capi.OUT_COMPACT, sample_chunks=spp)[0] for r in range(3)]
    assert np.array_equal(helpers.untile_numpy(np.concatenate(parts), width, height, 3).reshape(fb.shape), fb)


def test_sphere_scene_on_gpu(oracle, scenes, gpu_scenes, monkeypatch, node_cache):
    """Row N3 (scene.h:53-90): 488 spheres (487 of them sphere slots of the BVH), Lambertian / Metal / Dielectric, both
    kernels, whole pixels and sample chunks."""
    rgb, ids, st, _ = _compare(oracle, scenes, gpu_scenes, "random_spheres", 320, 180, 4, camera=scenes.SPHERES_CAMERA)
    _, s = gpu_scenes("random_spheres")
    _, cam = scenes.make_camera(320, **scenes.SPHERES_CAMERA)
    rgbc, idsc, stc = s.render(cam, 320, 180, 4, want_ids=True, sample_chunks=4)
    assert np.array_equal(idsc, ids) and stc.rays == st.rays and np.allclose(rgbc, rgb, rtol=3e-7, atol=1e-7)
    monkeypatch.setenv("CRT_KERNEL", "1")
    rgbp, idsp, stp = s.render(cam, 320, 180, 4, want_ids=True)
    monkeypatch.delenv("CRT_KERNEL", raising=False)
    assert np.array_equal(rgbp, rgb) and np.array_equal(idsp, ids) and stp.rays == st.rays


@pytest.mark.parametrize("name", ["cat", "cats"])
def test_gpu_lbvh_builder_same_ranks_same_image(oracle, capi, scenes, name):
    """Row N1: ranks and tree built on the device (crt_lbvh.hip).  The ranks must be the host's (= the oracle's leaf
    order); the tree is another one, the rendered result may not differ in a single bit."""
    desc = getattr(scenes, name)()
    host = scenes.build(desc, 0)
    gpu = scenes.build(desc, 0, capi.BUILD_GPU_LBVH)
    ih, ig = host.info(), gpu.info()
    assert (ih.builder, ig.builder) == (0, 1) and ig.num_triangles == ih.num_triangles and 0 < ig.bvh_depth <= 32
    assert np.array_equal(host.ranks(), gpu.ranks())
    assert np.allclose(np.array(ig.bounds_min), np.array(ih.bounds_min)) and np.allclose(np.array(ig.bounds_max), np.array(ih.bounds_max))
    width, height, spp = 320, 180, 4
    params, cam = scenes.make_camera(width)
    rgb_h, ids_h, st_h = host.render(cam, width, height, spp, want_ids=True)
    rgb_g, ids_g, st_g = gpu.render(cam, width, height, spp, want_ids=True)
    assert np.array_equal(ids_g, ids_h) and np.array_equal(rgb_g, rgb_h) and st_g.rays == st_h.rays
    fb, rids, _, rst = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp)
    assert np.array_equal(ids_g, rids) and np.array_equal(rgb_g.reshape(fb.shape), fb) and st_g.rays == rst.rays


def test_gpu_lbvh_builder_on_the_1m_triangle_scene(capi, scenes, node_cache):
    desc = scenes.cat_grid()
    host = scenes.build(desc, 0, 0)
    gpu = scenes.build(desc, 0, capi.BUILD_GPU_LBVH)
    ig = gpu.info()
    assert host.info().builder == 0 and ig.builder == 1 and ig.num_triangles == 999912 and ig.bvh_depth <= 32
    assert np.array_equal(host.ranks(), gpu.ranks())
    print("build: host SAH %.0f ms, GPU LBVH %.2f ms (depth %d vs %d)" % (host.info().build_ms, ig.build_ms, host.info().bvh_depth, ig.bvh_depth))
    assert ig.build_ms < 100.0
    _, cam = scenes.make_camera(480, **scenes.GRID_CAMERA)
    rgb_h, ids_h, st_h = host.render(cam, 480, 270, 2, want_ids=True)
    rgb_g, ids_g, st_g = gpu.render(cam, 480, 270, 2, want_ids=True)
    assert np.array_equal(ids_g, ids_h) and np.array_equal(rgb_g, rgb_h) and st_g.rays == st_h.rays


@pytest.mark.parametrize("name", ["cat", "cats"])
def test_gpu_sah_builder_same_ranks_same_image_same_cost(oracle, capi, scenes, name):
    """Row N1, second builder (crt_gsah.hip): the host builder's binned SAH run on the device.  Ranks are the host's, the frame
    is the oracle's bit for bit, and the tree is as good as the host's by the builders' own measure (surface-area cost)."""
    desc = getattr(scenes, name)()
    host = scenes.build(desc, 0, 0)
    gpu = scenes.build(desc, 0, capi.BUILD_GPU_SAH)
    ih, ig = host.info(), gpu.info()
    assert (ih.builder, ig.builder) == (0, 2) and ig.num_triangles == ih.num_triangles and 0 < ig.bvh_depth <= 32
    assert np.array_equal(host.ranks(), gpu.ranks())
    assert np.allclose(np.array(ig.bounds_min), np.array(ih.bounds_min)) and np.allclose(np.array(ig.bounds_max), np.array(ih.bounds_max))
    assert 0 < ig.sah_cost <= 1.02 * ih.sah_cost and abs(int(ig.num_nodes) - int(ih.num_nodes)) <= 0.05 * ih.num_nodes
    width, height, spp = 320, 180, 4
    params, cam = scenes.make_camera(width)
    rgb_h, ids_h, st_h = host.render(cam, width, height, spp, want_ids=True)
    rgb_g, ids_g, st_g = gpu.render(cam, width, height, spp, want_ids=True)
    assert np.array_equal(ids_g, ids_h) and np.array_equal(rgb_g, rgb_h) and st_g.rays == st_h.rays
    fb, rids, _, rst = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp)
    assert np.array_equal(ids_g, rids) and np.array_equal(rgb_g.reshape(fb.shape), fb) and st_g.rays == rst.rays


def test_gpu_sah_builder_on_the_1m_triangle_scene(capi, scenes, node_cache):
    """999,912 triangles: twenty levels of grid-wide kernels, then ~30 k small nodes finished by one wave each.  The tree must
    cost what the host's costs (within 5 % of the SAH tree's frame rate, built in <= 30 ms), and
    crt_scene_commit picks this builder on its own for a scene of this size."""
    desc = scenes.cat_grid()
    host = scenes.build(desc, 0, 0)
    gpu = scenes.build(desc, 0, capi.BUILD_GPU_SAH)
    gpu = scenes.build(desc, 0, capi.BUILD_GPU_SAH)  # (the first build in a process also loads the code object)
    auto = scenes.build(desc, 0)
    ih, ig = host.info(), gpu.info()
    assert (ih.builder, ig.builder, auto.info().builder) == (0, 2, 2) and ig.num_triangles == 999912 and ig.bvh_depth <= 32
    assert np.array_equal(host.ranks(), gpu.ranks())
    print("build: host SAH %.0f ms (cost %.3f, depth %d), GPU SAH %.2f ms (cost %.3f, depth %d)" % (ih.build_ms, ih.sah_cost, ih.bvh_depth, ig.build_ms, ig.sah_cost, ig.bvh_depth))
    assert ig.build_ms < 30.0 and ig.sah_cost <= 1.02 * ih.sah_cost
    _, cam = scenes.make_camera(480, **scenes.GRID_CAMERA)
    rgb_h, ids_h, st_h = host.render(cam, 480, 270, 2, want_ids=True)
    rgb_g, ids_g, st_g = gpu.render(cam, 480, 270, 2, want_ids=True, flags=capi.COUNT_WORK)
    st_hw = host.render(cam, 480, 270, 2, flags=capi.COUNT_WORK)[2]
    assert np.array_equal(ids_g, ids_h) and np.array_equal(rgb_g, rgb_h) and st_g.rays == st_h.rays
    assert st_g.nodes <= 1.03 * st_hw.nodes and st_g.tri_tests <= 1.03 * st_hw.tri_tests


def _sliver_scene(scenes, capi, n):
    """n nested, strongly overlapping slivers (tests/test_host.py: test_depth_bound_on_degenerate_input): SAH alone would build a
    chain; the depth bound has to force object medians."""
    d = scenes._base()
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.7, 0.3, 0.16), param=0.0, texture=None))
    scenes._add_ref_spheres(d)
    k = np.arange(n, dtype=np.float32)
    v = np.zeros((3 * n, 3), np.float32)
    v[0::3] = np.stack([0 * k - 1.0, 0 * k + 0.2, 0 * k], 1)
    v[1::3] = np.stack([1.0001 ** k - 1.0, 0 * k + 0.2, 0 * k], 1)
    v[2::3] = np.stack([0 * k - 1.0, 1.0001 ** k + 0.2, 1e-3 * k], 1)
    d["meshes"].append(dict(vertices=v, indices=np.arange(3 * n, dtype=np.uint32).reshape(n, 3), material=0))
    return d


@pytest.mark.parametrize("n", [2, 9, 63, 64, 65, 129, 1000])
def test_gpu_sah_builder_sizes_around_one_wave(oracle, scenes, capi, n):
    """Trees on both sides of the 64 triangles that one wave finishes on its own (n <= 64: no level kernels at all; 65: one level
    and two small nodes), against the oracle."""
    desc = scenes.terrain(2)
    me = desc["meshes"][0]
    rng = np.random.default_rng(n)
    c = rng.uniform(-1.5, 1.5, size=(n, 1, 3)).astype(np.float32) * np.array([1.0, 0.3, 1.0], np.float32) + np.array([0.0, 1.0, 0.0], np.float32)
    v = (c + rng.uniform(-0.4, 0.4, size=(n, 3, 3)).astype(np.float32)).reshape(3 * n, 3)
    desc["meshes"] = [dict(vertices=v, indices=np.arange(3 * n, dtype=np.uint32).reshape(n, 3), material=me["material"])]
    s = scenes.build(desc, 0, capi.BUILD_GPU_SAH)
    host = scenes.build(desc, 0, 0)
    assert s.info().builder == 2 and s.info().bvh_depth <= 32 and s.info().sah_cost <= 1.05 * host.info().sah_cost + 1e-6
    assert np.array_equal(s.ranks(), host.ranks())
    width, height, spp = 128, 72, 2
    params, cam = scenes.make_camera(width)
    rgb, ids, st = s.render(cam, width, height, spp, want_ids=True)
    fb, rids, _, rst = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp)
    assert np.array_equal(ids, rids) and np.array_equal(rgb.reshape(fb.shape), fb) and st.rays == rst.rays
    assert (ids >= 0).any() and (n < 9 or ((ids >= 0) & (ids < n)).any())


@pytest.mark.parametrize("kind", ["random", "coincident", "line"])
def test_gpu_sah_builder_on_awkward_large_inputs(oracle, scenes, capi, kind):
    """Inputs of 40,000 triangles (more than CRT_BUILD_AUTO's limit, so crt_scene_commit itself sends them to the device): a random
    soup, 40,000 triangles whose boxes share ONE centre (no plane separates anything: medians all the way down, in both tiers), and triangles
    strung along a line (two axes without extent).  Ranks are the host's, the tree fits the stack, the frame is the oracle's."""
    n = 40000
    rng = np.random.default_rng(7)
    d = scenes._base()
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.7, 0.3, 0.16), param=0.0, texture=None))
    scenes._add_ref_spheres(d)
    if kind == "random":
        c = rng.uniform(-2.0, 2.0, size=(n, 1, 3)).astype(np.float32) * np.array([1.0, 0.4, 1.0], np.float32) + np.array([0.0, 1.2, 0.0], np.float32)
        v = c + rng.uniform(-0.05, 0.05, size=(n, 3, 3)).astype(np.float32)
    elif kind == "coincident":
        # boxes of every size around ONE centre (extents on a 2^-10 grid, so the centres are equal bit for bit), triangles that
        # cross each other rather than coincide (coincident ones would all tie: 40,000 candidates per fold)
        e = (rng.integers(64, 1024, size=(n, 3)) / 1024.0).astype(np.float32)
        f = (rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32) * e)
        c0 = np.array([0.0, 1.25, 0.0], np.float32)
        v = np.stack([np.stack([-e[:, 0], -e[:, 1], f[:, 2]], -1), np.stack([e[:, 0], f[:, 1], -e[:, 2]], -1), np.stack([f[:, 0], e[:, 1], e[:, 2]], -1)], 1) + c0
    else:
        x = np.linspace(-3.0, 3.0, n, dtype=np.float32)
        v = np.stack([np.stack([x, 0 * x + 0.5, 0 * x], -1), np.stack([x + 0.01, 0 * x + 0.5, 0 * x], -1), np.stack([x, 0 * x + 1.5, 0 * x], -1)], 1)
    d["meshes"].append(dict(vertices=np.ascontiguousarray(v.reshape(3 * n, 3), dtype=np.float32), indices=np.arange(3 * n, dtype=np.uint32).reshape(n, 3), material=0))
    s = scenes.build(d, 0)
    host = scenes.build(d, 0, 0)
    i = s.info()
    assert i.builder == 2 and host.info().builder == 0 and 0 < i.bvh_depth <= 32 and i.num_triangles == n
    assert np.array_equal(s.ranks(), host.ranks())
    width, height, spp = 96, 54, 1
    params, cam = scenes.make_camera(width)
    rgb, ids, st = s.render(cam, width, height, spp, want_ids=True)
    rgb_h, ids_h, st_h = host.render(cam, width, height, spp, want_ids=True)
    assert np.array_equal(ids, ids_h) and np.array_equal(rgb, rgb_h) and st.rays == st_h.rays
    fb, rids, _, rst = oracle.RefScene(d).render(oracle.make_camera(params, width), width, height, spp)
    assert np.array_equal(ids, rids) and np.array_equal(rgb.reshape(fb.shape), fb) and st.rays == rst.rays
    assert ((ids >= 0) & (ids < n)).any()


def test_thousands_of_exactly_coincident_triangles(oracle, scenes, capi, node_cache):
    """4000 exact copies of each of three triangles: every hit is a tie among 4000 (8000 on the shared edge) candidates with equal
    t, and the reference's winner is the last of them in leaf order.  The fold finds it with one walk for the largest rank
    (crt_traverse.h: refold); candidate by candidate this frame took minutes.  Ids, image and ray count are the oracle's."""
    import time
    from tests.test_host import _stacked_copies
    copies = 4000
    desc = _stacked_copies(scenes, capi, copies)
    s = scenes.build(desc, 0)
    width, height, spp = 96, 54, 2
    params, cam = scenes.make_camera(width)
    t0 = time.perf_counter()
    rgb, ids, st = s.render(cam, width, height, spp, want_ids=True)
    seconds = time.perf_counter() - t0
    fb, rids, _, rst = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp)
    assert np.array_equal(ids, rids) and np.array_equal(rgb.reshape(fb.shape), fb) and st.rays == rst.rays
    assert ((ids >= 0) & (ids < 3 * copies)).sum() > 100 and st.tie_refolds > 100 and seconds < 5.0


def test_gpu_sah_builder_with_host_copies(oracle, scenes, capi, monkeypatch):
    """CRT_BUILD_HOST_COPIES=1 (a debugging switch): slots and normals come down to the host and go up again as before the
    replica adopted the builder's buffers; same scene, same frame."""
    desc = scenes.cats()
    adopted = scenes.build(desc, 0, capi.BUILD_GPU_SAH)
    monkeypatch.setenv("CRT_BUILD_HOST_COPIES", "1")
    copied = scenes.build(desc, 0, capi.BUILD_GPU_SAH)
    monkeypatch.delenv("CRT_BUILD_HOST_COPIES")
    assert adopted.info().builder == 2 and copied.info().builder == 2 and adopted.info().triangle_bytes == copied.info().triangle_bytes
    _, cam = scenes.make_camera(160)
    a, b = adopted.render(cam, 160, 90, 2, want_ids=True), copied.render(cam, 160, 90, 2, want_ids=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2].rays == b[2].rays


@pytest.mark.parametrize("n", [40, 5000])
def test_gpu_sah_builder_depth_bound_on_degenerate_input(oracle, scenes, capi, n):
    """Nested slivers: the SAH plane peels one sliver off at a time until the depth bound forces medians -- in the one-wave
    sub-tree builder (40) and in the level kernels (5000).  The tree stays within the traversal stack and renders the oracle's frame."""
    desc = _sliver_scene(scenes, capi, n)
    s = scenes.build(desc, 0, capi.BUILD_GPU_SAH)
    assert s.info().builder == 2 and 0 < s.info().bvh_depth <= 32
    width, height, spp = 96, 54, 1
    params, cam = scenes.make_camera(width, lookfrom=(0.3, 1.2, 4.0), lookat=(-0.4, 0.8, 0.0), vfov=40.0)
    rgb, ids, st = s.render(cam, width, height, spp, want_ids=True)
    fb, rids, _, rst = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp)
    assert np.array_equal(ids, rids) and np.array_equal(rgb.reshape(fb.shape), fb) and st.rays == rst.rays
    assert ((ids >= 0) & (ids < n)).any()


def test_randomised_configurations_against_oracle(oracle, scenes, gpu_scenes, capi, node_cache):
    """A sweep over odd frame sizes, seeds, depths, sample windows, partitions and chunk counts: every one bit-exact against
    the oracle (chunked runs against the oracle's chunk sums combined in chunk order)."""
    rng = np.random.default_rng(20240611)
    names = ["cat", "cats", "cat_checker", "random_spheres"]
    for trial in range(14):
        name = names[trial % len(names)]
        desc, s = gpu_scenes(name)
        width = int(rng.integers(9, 140))
        height = max(1, int(width / (16 / 9)))          # the camera's own height (camera.h:27)
        height = int(rng.integers(1, height + 1))       # or fewer rows
        spp = int(rng.choice([1, 2, 3, 4, 8]))
        kw = dict(max_depth=int(rng.integers(1, 12)), seed=int(rng.integers(0, 2 ** 40)), sample_offset=int(rng.integers(0, 50)))
        kw["ids_sample"] = kw["sample_offset"] + int(rng.integers(0, spp))
        camera = scenes.SPHERES_CAMERA if name == "random_spheres" else {}
        params, cam = scenes.make_camera(width, **camera)
        rs = oracle.RefScene(desc)
        ocam = oracle.make_camera(params, width)
        fb, rids, acc, rst = rs.render(ocam, width, height, spp, want_accum=True, **kw)
        rgb, ids, st = s.render(cam, width, height, spp, want_ids=True, **kw)
        assert np.array_equal(ids, rids) and np.array_equal(rgb.reshape(fb.shape), fb) and st.rays == rst.rays, (trial, name, width, height, spp, kw)
        # tile partition, compact, gathered on the host
        ranks = int(rng.integers(2, 6))
        parts = [s.render(cam, width, height, spp, tile_rank=r, tile_ranks=ranks, flags=capi.OUT_COMPACT, **kw)[0] for r in range(ranks)]
        assert np.array_equal(helpers.untile_numpy(np.concatenate(parts), width, height, ranks), fb), (trial, "tiles", ranks)
        # sample chunks
        chunks = [c for c in (2, 3, 4, 5, 6, 7, 8, 9, 11) if spp % c == 0 and c <= spp]
        if chunks:
            n = int(rng.choice(chunks))
            total = None
            for c in range(n):
                ckw = dict(kw)
                ckw["sample_offset"] = kw["sample_offset"] + c * (spp // n)
                _, _, a, _ = rs.render(ocam, width, height, spp // n, want_ids=False, want_accum=True, **ckw)
                total = a.copy() if total is None else (total + a).astype(np.float32)
            accn, idsn, stn = s.render(cam, width, height, spp, want_ids=True, flags=capi.OUT_ACCUM, sample_chunks=n, **kw)
            assert np.array_equal(accn.reshape(total.shape), total) and np.array_equal(idsn, rids) and stn.rays == rst.rays, (trial, "chunks", n)


def test_pipelined_tile_gather_over_rccl_single_rank(scenes, gpu_scenes, capi):
    """dist.TileFrame's pipelined path on the real backend: a 1-rank nccl (= RCCL) group, three frames with two all-gathers
    in flight; every frame that comes out equals the direct render of its seed, bit for bit."""
    import socket
    import torch
    import torch.distributed as dist
    from cudaraytracer_amd.dist import TileFrame
    desc, s = gpu_scenes("cats")
    W, H, spp = 200, 113, 4
    _, cam = scenes.make_camera(W)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    dev = torch.device("cuda", 0)
    kw = {}
    if os.environ.get("CRT_TEST_PRIORITY_STREAM") == "1":   # the collectives on a high-priority stream (diagnostic runs)
        kw["pg_options"] = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev, **kw)
    try:
        for in_flight in (True, 3):   # two frames in flight (one GPU's default) and three (bench.py's default when ranks share a frame)
            f = TileFrame(W, H, 1, 0, dev, sample_chunks=2, pipelined=in_flight, force_collective=True)
            seeds, got = (3, 5, 7, 11, 13, 1984, 2, 17), []   # eight frames over four or five buffer sets: every set is used at least once more
            for seed in seeds:
                f.render_local(s, cam, spp, 10, seed)
                f.gather_async()
                with torch.cuda.stream(f.slots.epilogue):   # the frame is complete on the epilogue stream (drain() hands it to the caller's)
                    got.append(f.frame.clone())
            last = f.drain().clone()
            torch.cuda.synchronize()
            for seed, frame in zip(seeds, got):
                want = s.render(cam, W, H, spp, seed=seed, sample_chunks=2)[0]
                assert np.array_equal(frame.cpu().numpy().reshape(want.shape), want), (in_flight, seed)
            assert np.array_equal(last.cpu().numpy().reshape(want.shape), want)
        # and without a collective: two frames in flight straight into the frame buffers
        g = TileFrame(W, H, 1, 0, dev, sample_chunks=2, pipelined=True)
        for seed in seeds:
            g.render_local(s, cam, spp, 10, seed)
            g.gather_async()
        last = g.drain().clone()
        torch.cuda.synchronize()
        assert np.array_equal(last.cpu().numpy().reshape(want.shape), want)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("spp,n", [(64, 64), (128, 128), (256, 32), (30, 6), (21, 3), (130, 130)])
def test_many_sample_chunks(oracle, scenes, gpu_scenes, capi, chunk_mode, spp, n):
    """Chunk counts at and above the 64 units of a work block (a pixel's chunks then span several blocks and several waves),
    on a ragged frame; raw sums against the oracle's per-chunk sums added in chunk order, ids and ray count exact."""
    desc, s = gpu_scenes("cats")
    width, height = 43, 19
    params, cam = scenes.make_camera(width)
    rs = oracle.RefScene(desc)
    ocam = oracle.make_camera(params, width)
    total = None
    rays = 0
    for c in range(n):
        _, _, acc, st = rs.render(ocam, width, height, spp // n, sample_offset=c * (spp // n), want_ids=False, want_accum=True)
        total = acc.copy() if total is None else (total + acc).astype(np.float32)
        rays += st.rays
    _, rids, _, _ = rs.render(ocam, width, height, 1, sample_offset=spp - 1, ids_sample=spp - 1, want_accum=True)
    acc, ids, st = s.render(cam, width, height, spp, want_ids=True, ids_sample=spp - 1, flags=capi.OUT_ACCUM, sample_chunks=n)
    assert st.rays == rays and np.array_equal(ids, rids)
    assert np.array_equal(acc.reshape(total.shape), total)


def _degenerate_scene(scenes, capi, kind):
    d = scenes._base()
    scenes._add_ref_spheres(d)                        # ground + light (flat list)
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.7, 0.3, 0.16), param=0.0, texture=None))
    mat = len(d["materials"]) - 1
    if kind == "spheres_only":                        # no triangles at all: no tree, root reference = none
        d["materials"].append(dict(type=capi.MAT_METAL, rgb=(0.8, 0.8, 0.8), param=0.1, texture=None))
        d["spheres"].append(dict(center=(0.0, 1.0, 0.0), radius=1.0, material=mat))
        d["spheres"].append(dict(center=(2.2, 0.7, 0.5), radius=0.7, material=len(d["materials"]) - 1))
    else:                                             # 1 triangle: the root is a leaf; 3: one node above two leaves
        n = 1 if kind == "one_triangle" else 3
        v = np.array([[-2.0, 0.0, -1.0], [2.0, 0.0, -1.0], [0.0, 3.0, 0.0], [-2.5, 0.2, 1.0], [-0.5, 0.2, 1.5], [-1.5, 2.0, 1.2],
                      [0.5, 0.1, 2.0], [2.5, 0.1, 1.5], [1.5, 1.8, 1.8]], dtype=np.float32)[: 3 * n]
        idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        d["meshes"].append(dict(vertices=v, indices=idx, material=mat))
    return d


@pytest.mark.parametrize("kind", ["spheres_only", "one_triangle", "three_triangles"])
def test_degenerate_trees(oracle, scenes, capi, node_cache, kind):
    """Scenes whose tree is missing (spheres only), a single leaf, or a single node: ids, image and ray count against the oracle."""
    desc = _degenerate_scene(scenes, capi, kind)
    s = scenes.build(desc, 0)
    width, height, spp = 96, 54, 3
    params, cam = scenes.make_camera(width)
    rgb, ids, st = s.render(cam, width, height, spp, want_ids=True, ids_sample=1)
    fb, rids, _, rst = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp, ids_sample=1)
    assert np.array_equal(ids, rids) and np.array_equal(rgb, fb) and st.rays == rst.rays
    assert (ids >= 0).any()


@pytest.mark.parametrize("builder", ["lbvh", "sah"])
@pytest.mark.parametrize("kind", ["spheres_only", "one_triangle", "three_triangles"])
def test_degenerate_trees_from_the_gpu_builder(oracle, scenes, capi, kind, builder):
    """The same degenerate scenes committed through the GPU builders (a scene without triangles has nothing to build there)."""
    desc = _degenerate_scene(scenes, capi, kind)
    s = scenes.build(desc, 0, flags=capi.BUILD_GPU_LBVH if builder == "lbvh" else capi.BUILD_GPU_SAH)
    width, height, spp = 96, 54, 2
    params, cam = scenes.make_camera(width)
    rgb, ids, st = s.render(cam, width, height, spp, want_ids=True)
    fb, rids, _, rst = oracle.RefScene(desc).render(oracle.make_camera(params, width), width, height, spp)
    assert np.array_equal(ids, rids) and np.array_equal(rgb, fb) and st.rays == rst.rays


@pytest.mark.parametrize("camera", [dict(defocus_angle=0.0), dict(member_defocus_angle=0.0), dict(member_defocus_angle=-1.0, defocus_angle=4.0),
                                    dict(defocus_angle=4.0, vfov=60.0, lookfrom=(-3.0, 2.0, 7.0), lookat=(0.0, 1.0, 0.0)),
                                    dict(vfov=10.0, aspect=1.0, lookfrom=(0.0, 9.0, 0.5), lookat=(0.0, 0.0, 0.0), vup=(0.0, 0.0, -1.0))])
def test_other_cameras(oracle, scenes, gpu_scenes, node_cache, camera):
    """A constructor defocus_angle of 0 (the disk has radius 0 but is still sampled: the parameter shadows the member, which
    stays 0.6, camera.h:20,55,94); the MEMBER Camera::defocus_angle written to 0 or below after construction, which is what
    takes get_ray's pinhole branch (camera.h:94: the centre is the origin and no disk draw is made -- another shape of the
    camera stage's random stream); a wide lens (most disk draws are retried at least once somewhere in a wave); a top view
    with another up vector and a square aspect."""
    _compare(oracle, scenes, gpu_scenes, "cats", 120, 67 if "aspect" not in camera else 120, 3, camera=camera, ids_sample=2)


def test_dielectric_and_checker_meshes(oracle, scenes, capi, node_cache):
    """A glass cat beside a metal one on a checkered ground sphere: refraction / total reflection through triangle meshes
    (material.h:61-91 with triangle normals) and the checker's sign handling on a large sphere (texture.h:31-39)."""
    d = scenes._base()
    v, i = capi.load_obj(os.path.join(scenes.MODELS, "cat.obj"))
    v = v[:, :3].copy()
    d["materials"].append(dict(type=capi.MAT_DIELECTRIC, rgb=(1.0, 1.0, 1.0), param=1.5, texture=None))
    d["materials"].append(dict(type=capi.MAT_METAL, rgb=(0.9, 0.6, 0.2), param=0.3, texture=None))
    d["meshes"].append(dict(vertices=v, indices=i, material=0))
    d["meshes"].append(dict(vertices=v + np.array([2.5, 0.0, 1.0], np.float32), indices=i, material=1))
    d["textures"].append(dict(type=capi.TEX_CHECKER, a=(0.2, 0.3, 0.1), b=(0.9, 0.9, 0.9), scale=0.7))
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.5, 0.5, 0.5), param=0.0, texture=0))
    d["materials"].append(dict(type=capi.MAT_DIFFUSE_LIGHT, rgb=(15.0, 15.0, 15.0), param=0.0, texture=None))
    d["spheres"].append(dict(center=(0.0, -1000.0, -1.0), radius=1000.0, material=2))
    d["spheres"].append(dict(center=(3.0, 6.5, 3.0), radius=2.0, material=3))
    s = scenes.build(d, 0)
    width, height, spp = 160, 90, 4
    params, cam = scenes.make_camera(width)
    rgb, ids, st = s.render(cam, width, height, spp, want_ids=True, max_depth=12)
    fb, rids, _, rst = oracle.RefScene(d).render(oracle.make_camera(params, width), width, height, spp, max_depth=12)
    assert np.array_equal(ids, rids) and st.rays == rst.rays
    assert np.array_equal(rgb, fb), "PSNR %.1f dB" % helpers.psnr(rgb, fb)


def test_work_counters_equal_the_host_single_stepper(hostsim, scenes, gpu_scenes, capi, node_cache):
    """The CRT_COUNT_WORK counters price the roofline (bench.py: lane_ops): nodes fetched, triangle / sphere tests, shaded
    hits, path segments and tie refolds of a frame must equal the counts of the same per-lane code single-stepped on the
    host (tests/hostsim), whatever the wave schedule and the kernel variant."""
    desc, s = gpu_scenes("cats")
    W, H, spp = 200, 113, 3
    _, cam = scenes.make_camera(W)
    hs = hostsim.build(desc)
    want_rgb, want_ids, want = hostsim.render(hs, cam, W, H, spp)
    for chunks in (0, 1):          # whole pixels; (3 samples: no power-of-two chunking beyond 1)
        rgb, ids, st = s.render(cam, W, H, spp, want_ids=True, flags=capi.COUNT_WORK, sample_chunks=chunks)
        got = dict(rays=st.rays, nodes=st.nodes, tri_tests=st.tri_tests, sphere_tests=st.sphere_tests, tie_refolds=st.tie_refolds, shaded_hits=st.shaded_hits)
        assert got == {k: want[k] for k in got}, (got, want)
        assert np.array_equal(rgb, want_rgb) and np.array_equal(ids, want_ids)
    # chunked units walk the same rays: same counts
    st4 = s.render(cam, W, H, 4, flags=capi.COUNT_WORK, sample_chunks=4)[2]
    w4 = hostsim.render(hs, cam, W, H, 4)[2]
    assert (st4.rays, st4.nodes, st4.tri_tests, st4.sphere_tests, st4.shaded_hits) == (w4["rays"], w4["nodes"], w4["tri_tests"], w4["sphere_tests"], w4["shaded_hits"])


@pytest.mark.parametrize("node_cache_wide", ["0", "2", "4"])
def test_four_child_nodes_same_frames_half_the_fetches(oracle, hostsim, scenes, capi, monkeypatch, node_cache_wide):
    """CRT_BVH_WIDE=1 (crt_bvh.h: widen_bvh): the tree collapsed to four children per node is walked by kernels of their own;
    frames, ids and ray counts are the oracle's, the work counters the host single-stepper's for the same tree."""
    monkeypatch.setenv("CRT_BVH_WIDE", "1")
    monkeypatch.setenv("CRT_NODE_CACHE", node_cache_wide)
    for name, W, H, spp, chunks in (("cats", 200, 113, 3, 0), ("cats", 97, 45, 8, 8), ("random_spheres", 120, 68, 2, 2), ("cat_checker", 64, 36, 4, 2)):
        desc = getattr(scenes, name)()
        s = scenes.build(desc, 0)
        info = s.info()
        assert info.node_width == 4 and info.bvh_depth <= (info.binary_depth + 1) // 2 + 1 and info.node_bytes == 128 * info.num_nodes
        params, cam = scenes.make_camera(W)
        rgb, ids, st = s.render(cam, W, H, spp, want_ids=True, flags=capi.COUNT_WORK, sample_chunks=chunks)
        fb, rids, _, rst = oracle.RefScene(desc).render(oracle.make_camera(params, W), W, H, spp)
        if chunks == 0:
            assert np.array_equal(rgb, fb)
        assert np.array_equal(ids, rids) and st.rays == rst.rays
        hs = hostsim.build(desc)
        want_rgb, want_ids, want = hostsim.render(hs, cam, W, H, spp)
        got = dict(rays=st.rays, nodes=st.nodes, tri_tests=st.tri_tests, sphere_tests=st.sphere_tests, tie_refolds=st.tie_refolds, shaded_hits=st.shaded_hits)
        assert got == {k: want[k] for k in got}, (got, want)
        # the same scene with two children per node: the same frame bit for bit, about twice the node fetches
        monkeypatch.setenv("CRT_BVH_WIDE", "0")
        s2 = scenes.build(desc, 0)
        monkeypatch.setenv("CRT_BVH_WIDE", "1")
        rgb2, ids2, st2 = s2.render(cam, W, H, spp, want_ids=True, flags=capi.COUNT_WORK, sample_chunks=chunks)
        assert s2.info().node_width == 2 and np.array_equal(rgb, rgb2) and np.array_equal(ids, ids2)
        assert st.nodes < 0.75 * st2.nodes and st.tri_tests <= 1.1 * st2.tri_tests


@pytest.mark.parametrize("n", [64, 32])
def test_benchmarked_configuration_at_full_size(scenes, gpu_scenes, capi, n):
    """What bench.py times: cats, 1920x1080, 64 spp, sample_chunks = 64 (one sample per work unit; 32 was the default before
    and stays covered), frames alternating between two streams.  The raw sums of the chunked frame must equal, bit for bit,
    the chunk-order float32 sum of n whole-pixel renders of 64/n spp at sample_offset = c * 64/n (each of those is the
    oracle-checked sequential path), and four frames issued back to back on two alternating streams must all come out as
    that same frame.  With one sample per chunk the chunk-order sum IS the reference's sequential loop (renderer.h:26-32), so
    the timed frame must also equal the whole-pixel frame (sample_chunks = 1) bit for bit, raw sums and finished image."""
    import torch
    _, s = gpu_scenes("cats")
    W, H, spp = 1920, 1080, 64
    _, cam = scenes.make_camera(W)
    chunked = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
    st = s.render(cam, W, H, spp, flags=capi.OUT_ACCUM | capi.OUT_DEVICE, out_rgb=chunked.data_ptr(), sample_chunks=n)[2]
    total = torch.zeros_like(chunked)
    part = torch.empty_like(chunked)
    rays = 0
    for c in range(n):
        rays += s.render(cam, W, H, spp // n, sample_offset=c * (spp // n), flags=capi.OUT_ACCUM, out_rgb=part.data_ptr())[2].rays
        total = part.clone() if c == 0 else total + part
    torch.cuda.synchronize()
    assert rays == st.rays and torch.equal(chunked, total)
    # finished frame = those sums / spp, gamma (renderer.h:36-37, color.h:5-24)
    frame = torch.empty_like(chunked)
    s.render(cam, W, H, spp, out_rgb=frame.data_ptr(), sample_chunks=n)
    v = total / spp
    want = torch.where(v > 0, torch.sqrt(v), torch.zeros_like(v))
    torch.cuda.synchronize()
    assert torch.equal(frame, want)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.zeros_like(chunked) for _ in range(4)]
    for k in range(4):
        s.render(cam, W, H, spp, out_rgb=outs[k].data_ptr(), sample_chunks=n, want_stats=False, stream=streams[k % 2].cuda_stream,
                 flags=capi.OUT_DEVICE)
    for st_ in streams:
        st_.synchronize()
    for k in range(4):
        assert torch.equal(outs[k], frame), k
    if n == spp:
        whole = torch.empty_like(chunked)
        st1 = s.render(cam, W, H, spp, flags=capi.OUT_ACCUM | capi.OUT_DEVICE, out_rgb=whole.data_ptr(), sample_chunks=1)[2]
        torch.cuda.synchronize()
        assert st1.rays == st.rays and torch.equal(whole, chunked)
        s.render(cam, W, H, spp, out_rgb=whole.data_ptr(), sample_chunks=1, flags=capi.OUT_DEVICE)
        torch.cuda.synchronize()
        assert torch.equal(whole, frame)


def test_host_pointer_renders_on_two_streams_from_two_threads(scenes, gpu_scenes):
    """Host-pointer frames are staged through a device buffer; each stream has its own, so two threads rendering different
    seeds on two streams never see each other's pixels."""
    import threading
    import torch
    _, s0 = gpu_scenes("cats")
    W, H, spp = 320, 180, 4
    _, cam = scenes.make_camera(W)
    seeds = {0: (11, 13, 17, 19), 1: (23, 29, 31, 37)}
    want = {seed: s0.render(cam, W, H, spp, seed=seed, want_ids=True)[:2] for ss in seeds.values() for seed in ss}
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got, errors = {}, []

    def worker(t):
        try:
            for seed in seeds[t]:
                rgb, ids, _ = s0.render(cam, W, H, spp, seed=seed, want_ids=True, want_stats=(seed % 2 == 1), stream=streams[t].cuda_stream)
                got[seed] = (rgb, ids)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for seed, (rgb, ids) in want.items():
        assert np.array_equal(got[seed][0], rgb) and np.array_equal(got[seed][1], ids), seed


@pytest.mark.parametrize("workload", ["cats_4k", "cat_grid_4k"])
def test_full_size_properties_3840x2160(scenes, gpu_scenes, capi, workload):
    """BASELINE configs 4 and 5 at their own frame size (3840x2160: 518,400 tiles; with 32 sample chunks 2^24.9 work units
    and 3.2 GB of chunk slabs per stream), too slow for the oracle, so size-independent properties: bitwise run-to-run
    determinism, tile split over 8 ranks (compact lists + crt_untile) == single launch bitwise, sample-split ray counts add
    up, and the chunked frame agrees with the whole-pixel (sequential-sum, oracle-checked path) frame to float association."""
    import torch
    if workload == "cats_4k":
        _, s = gpu_scenes("cats")
        spp, chunks, cam_over = 32, 32, {}
    else:
        _, s = gpu_scenes("cat_grid")
        spp, chunks, cam_over = 16, 16, scenes.GRID_CAMERA
    W, H = 3840, 2160
    _, cam = scenes.make_camera(W, **cam_over)
    dev = capi.OUT_DEVICE
    a = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    st_a = s.render(cam, W, H, spp, out_rgb=a.data_ptr(), sample_chunks=chunks, flags=dev)[2]
    st_b = s.render(cam, W, H, spp, out_rgb=b.data_ptr(), sample_chunks=chunks, flags=dev)[2]
    torch.cuda.synchronize()
    assert torch.equal(a, b) and st_a.rays == st_b.rays
    ranks = 8
    n = capi.compact_floats(W, H, ranks)
    allc = torch.zeros(ranks * n, dtype=torch.float32, device="cuda")
    rays = 0
    for r in range(ranks):
        rays += s.render(cam, W, H, spp, tile_rank=r, tile_ranks=ranks, flags=capi.OUT_COMPACT | dev, out_rgb=allc[r * n:].data_ptr(), sample_chunks=chunks)[2].rays
    capi.untile(allc.data_ptr(), W, H, ranks, b.data_ptr())
    torch.cuda.synchronize()
    assert rays == st_a.rays and torch.equal(a, b)
    del allc
    half = spp // 2
    r0 = s.render(cam, W, H, half, sample_offset=0, flags=capi.OUT_ACCUM | dev, out_rgb=b.data_ptr(), sample_chunks=chunks // 2)[2].rays
    r1 = s.render(cam, W, H, half, sample_offset=half, flags=capi.OUT_ACCUM | dev, out_rgb=b.data_ptr(), sample_chunks=chunks // 2)[2].rays
    assert r0 + r1 == st_a.rays
    st_w = s.render(cam, W, H, spp, out_rgb=b.data_ptr(), flags=dev)[2]          # whole pixels: the reference's sequential sums
    torch.cuda.synchronize()
    assert st_w.rays == st_a.rays and torch.allclose(a, b, rtol=3e-7, atol=1e-7)
    img = a.cpu().numpy().reshape(H, W, 3)
    assert np.isfinite(img).all() and img.min() >= 0 and img.mean() > 0.01


def test_config5_at_its_own_sample_count_3840x2160x1024(scenes, gpu_scenes, capi):
    """BASELINE config 5 in full: cats at 3840 x 2160 with 1024 samples per pixel (8.5 G primary rays; half a second per frame).
    One sample per work unit -- 3.3 TB as slabs, so the frame goes through the waves' rings -- must equal the whole-pixel
    frame (the reference's sequential sum, renderer.h:26-32, the path checked against the oracle at small sizes) bit for
    bit; and the sample split the config names (8 ranks x 128 samples, raw sums added in rank order, then / spp and gamma)
    must account for every ray and agree with the full frame to float association (tolerance: 8e-6 relative on the
    gamma-encoded value, stated where it is asserted)."""
    import torch
    _, s = gpu_scenes("cats")
    W, H, spp, ranks = 3840, 2160, 1024, 8
    _, cam = scenes.make_camera(W)
    dev = capi.OUT_DEVICE
    a = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    st_a = s.render(cam, W, H, spp, out_rgb=a.data_ptr(), sample_chunks=spp, flags=dev)[2]
    st_w = s.render(cam, W, H, spp, out_rgb=b.data_ptr(), flags=dev)[2]
    torch.cuda.synchronize()
    assert st_a.rays == st_w.rays and torch.equal(a, b)
    print("4K x 1024 spp: %.0f ms with one sample per unit, %.0f ms with whole pixels, %.1f Grays/s" % (st_a.kernel_ms, st_w.kernel_ms, st_a.rays / st_a.kernel_ms / 1e6))
    acc = torch.zeros_like(a)
    part = torch.empty_like(a)
    rays = 0
    for r in range(ranks):
        rays += s.render(cam, W, H, spp // ranks, sample_offset=r * (spp // ranks), flags=capi.OUT_ACCUM | dev, out_rgb=part.data_ptr(), sample_chunks=64)[2].rays
        torch.cuda.synchronize()
        acc += part
    capi.finalize(acc.data_ptr(), W, H, spp, b.data_ptr())
    torch.cuda.synchronize()
    # 1024 positive terms added in another association: the sums differ by a few sqrt(n) * 2^-24 relative (worst case n * 2^-24),
    # halved by the gamma's square root; measured 2e-6 at most
    rel = ((a - b).abs() / b.abs().clamp_min(1e-3)).max().item()
    print("sample split 8 x 128 against the sequential sum: largest relative difference %.2e" % rel)
    assert rays == st_a.rays and torch.allclose(a, b, rtol=8e-6, atol=1e-6)


_TWO_PART_OBJ = """# two objects, two materials, one of them emissive (row N4: crt_load_obj_ex + .mtl records)
mtllib parts.mtl
o body
v -1.5 0.0 -1.0
v 1.5 0.0 -1.0
v 0.0 2.4 -0.2
v 0.0 0.0 1.6
vn 0 0 1
usemtl clay
f 1//1 2//1 3//1
f 1//1 3//1 4//1
f 2//1 4//1 3//1
o lamp
v -1.0 3.2 -1.0
v 1.0 3.2 -1.0
v 1.0 3.2 1.0
v -1.0 3.2 1.0
usemtl glow
f 5//1 6//1 7//1 8//1
"""
_TWO_PART_MTL = """newmtl clay
Kd 0.8 0.25 0.1
Ns 10
newmtl glow
Kd 0.1 0.1 0.1
Ke 12.0 11.0 9.0
illum 2
"""


def test_obj_groups_and_mtl_materials_render_like_the_oracle(oracle, scenes, capi, tmp_path, node_cache):
    """Row N4: a two-object, two-usemtl OBJ whose .mtl has an emissive record, through crt_load_obj_ex, mapped to materials
    as Scene::materials_from_mtl does (DiffuseLight(Ke) where Ke > 0, else Lambertian(Kd)), rendered on the GPU and by the
    oracle fed the same groups.  The quad is fan-triangulated; indices refer to the one joined vertex array."""
    (tmp_path / "parts.obj").write_text(_TWO_PART_OBJ)
    (tmp_path / "parts.mtl").write_text(_TWO_PART_MTL)
    v, idx, tri_group, groups = capi.load_obj_groups(str(tmp_path / "parts.obj"))
    assert [(g["object"], g["material"], g["num_triangles"]) for g in groups] == [("body", "clay", 3), ("lamp", "glow", 2)]
    assert np.allclose(groups[1]["Ke"], (12.0, 11.0, 9.0)) and np.allclose(groups[0]["Kd"], (0.8, 0.25, 0.1)) and idx.shape == (5, 3)
    assert np.array_equal(tri_group, [0, 0, 0, 1, 1])
    d = scenes._base()
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.7, 0.3, 0.16), param=0.0, texture=None))   # scene.h:147-148 (unused here, as in Scene)
    for g in groups:
        emits = any(c > 0 for c in g["Ke"])
        d["materials"].append(dict(type=capi.MAT_DIFFUSE_LIGHT if emits else capi.MAT_LAMBERTIAN, rgb=tuple(g["Ke"] if emits else g["Kd"]), param=0.0, texture=None))
        d["meshes"].append(dict(vertices=v[:, :3].copy(), indices=idx[g["first_triangle"]: g["first_triangle"] + g["num_triangles"]],
                                material=len(d["materials"]) - 1))
    scenes._add_ref_spheres(d)
    s = scenes.build(d, 0)
    W, H, spp = 160, 90, 4
    params, cam = scenes.make_camera(W)
    rgb, ids, st = s.render(cam, W, H, spp, want_ids=True)
    fb, rids, _, rst = oracle.RefScene(d).render(oracle.make_camera(params, W), W, H, spp)
    assert np.array_equal(ids, rids) and np.array_equal(rgb, fb) and st.rays == rst.rays
    assert set(np.unique(ids)) >= {0, 3}            # the clay body and the lamp are both seen directly
    # and the same file through the C++ surface (Scene::materials_from_mtl = true): examples/crt_demo --mtl
    import subprocess
    demo = os.path.join(helpers.REPO, "examples", "crt_demo")
    r = subprocess.run([demo, str(tmp_path / "parts.obj"), str(W), str(H), str(spp), "--mtl"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(tmp_path / "output.ppm", "rb").read()
    header = ("P6\n%d %d\n255\n" % (W, H)).encode()
    img = np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(H, W, 3)
    assert np.array_equal(img, np.clip(fb * np.float32(255.0), 0, 255).astype(np.uint8))


def test_fuzz_findings_round_3_sphere_hits_outside_the_spheres_box(oracle, scenes, gpu_scenes, monkeypatch):
    """The two frames of tools/fuzz_parity.py's round-3 campaign that differed from the oracle by one ray each: a sphere 'hit' ~1e-4
    outside the silhouette (Sphere::hit's discriminant cancels) at a point where the silhouette touches the sphere's own unpadded
    box, which the reference's box test has already turned away.  The kernels take such a hit provisionally and settle it on the
    careful path with the literal test (crt_core.h: sphere_t, crt_traverse.h: refold); both kernels, at the frames' own parameters."""
    desc, s = gpu_scenes("random_spheres")
    rs = oracle.RefScene(desc)
    for w, h, spp, extra, kw in ((191, 39, 6, {}, dict(max_depth=4, seed=813596887249, sample_offset=40, ids_sample=45)),
                                 (179, 27, 8, dict(defocus_angle=0.0), dict(max_depth=7, seed=775945998277, sample_offset=33, ids_sample=33))):
        cam_kw = dict(scenes.SPHERES_CAMERA)
        cam_kw.update(extra)
        params, cam = scenes.make_camera(w, **cam_kw)
        fb, rids, _, rst = rs.render(oracle.make_camera(params, w), w, h, spp, **kw)
        for kernel in (None, "1"):
            if kernel:
                monkeypatch.setenv("CRT_KERNEL", kernel)
            rgb, ids, st = s.render(cam, w, h, spp, want_ids=True, **kw)
            monkeypatch.delenv("CRT_KERNEL", raising=False)
            assert st.rays == rst.rays and np.array_equal(ids, rids) and np.array_equal(rgb.reshape(fb.shape), fb), (w, kernel)


def test_spheres_hovering_above_a_floor_seen_from_below(oracle, scenes, capi):
    """tests/test_host.py::test_a_sphere_root_before_its_own_box_is_not_pruned as frames on the GPU: 15 spheres 2e-4 above a floor,
    cameras below the floor looking up (far away: the spheres' computed roots are noisy enough to come out in front of the floor)."""
    from tests.test_host import _hovering_spheres
    desc, _ = _hovering_spheres(scenes, capi, 2e-4)
    s = scenes.build(desc, 0)
    assert s.info().num_spheres == 15 and s.info().sphere_bytes == 0       # all of them in the tree
    rs = oracle.RefScene(desc)
    differs_from_the_floor = 0
    for lookfrom, vfov in (((3.0, -40.0, 9.0), 8.0), ((-25.0, -30.0, 4.0), 9.0), ((0.5, -80.0, 0.3), 4.0)):
        params, cam = scenes.make_camera(320, lookfrom=lookfrom, lookat=(0.0, 0.0, 0.0), vfov=vfov, defocus_angle=0.0, background=(0.6, 0.7, 0.9))
        fb, rids, _, rst = rs.render(oracle.make_camera(params, 320), 320, 180, 2, max_depth=3)
        rgb, ids, st = s.render(cam, 320, 180, 2, want_ids=True, max_depth=3)
        assert st.rays == rst.rays and np.array_equal(ids, rids) and np.array_equal(rgb.reshape(fb.shape), fb), lookfrom
        differs_from_the_floor += int((rids >= 2).sum())
    assert differs_from_the_floor > 0                                       # spheres show through the floor, as in the reference
