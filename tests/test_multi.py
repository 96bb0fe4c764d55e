"""Multi-GPU behind the C ABI (crt_scene_commit_devices / crt_render_multi): one process drives n devices.  On the one-GPU
test box the n logical ranks all map to device 0 (the exchange is then peer copies, since an RCCL communicator refuses
duplicate devices) plus a 1-device RCCL communicator, which sends the frame through ncclAllGather / ncclReduce for real."""
import os
import subprocess

import numpy as np
import pytest

from tests import helpers


def test_tile_bookkeeping_on_the_host(capi):
    """crt_compact_floats and the tile -> (rank, slot) dealing that crt_render_multi, crt_untile and dist.py share: every
    pixel of a ragged frame is owned by exactly one rank, and the compact lists of all ranks have the same length."""
    for (w, h, n) in [(1920, 1080, 8), (203, 117, 3), (8, 8, 5), (65, 9, 2), (3840, 2160, 8)]:
        tiles = ((w + 7) // 8) * ((h + 7) // 8)
        per = (tiles + n - 1) // n
        assert capi.compact_floats(w, h, n) == per * 64 * 3
        if w * h > 100000:
            continue
        m = helpers.tile_index_map(w, h, n)
        owned = m[m >= 0]
        assert owned.size == w * h and np.array_equal(np.sort(owned), np.arange(w * h))
        frame = np.arange(w * h * 3, dtype=np.float32).reshape(h, w, 3)
        parts = np.concatenate([helpers.to_compact(frame, r, n) for r in range(n)])
        assert np.array_equal(helpers.untile_numpy(parts, w, h, n), frame)


def test_multi_entry_points_fail_loudly_without_devices(capi, scenes):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    s = scenes.build(scenes.cat(), capi.DEVICE_NONE)
    with pytest.raises(capi.CrtError) as e:
        s.commit_devices([0, 0])
    assert e.value.code in (capi.CRT_ERR_NO_DEVICE, capi.CRT_ERR_HIP)
    _, cam = scenes.make_camera(64)
    with pytest.raises(capi.CrtError):
        s.render(cam, 64, 36, 1, multi=capi.MULTI_TILE)


def _commit(scenes, capi, desc, devices):
    s = capi.Scene()
    for t in desc["textures"]:
        s.add_texture(t["type"], t["a"], t.get("b"), t.get("scale", 1.0))
    for m in desc["materials"]:
        s.add_material(m["type"], m["rgb"], m["param"], capi.NO_TEXTURE if m["texture"] is None else m["texture"])
    for me in desc["meshes"]:
        s.add_mesh(me["vertices"], me["indices"], me["material"])
    for sp in desc["spheres"]:
        s.add_sphere(sp["center"], sp["radius"], sp["material"])
    s.commit_devices(devices)
    return s


@pytest.mark.gpu
def test_replicas_of_a_device_built_scene_copy_from_the_first(capi, scenes):
    """A scene large enough for crt_scene_commit_devices to build its tree on devices[0] (crt_gsah.hip) has no host copy of its
    slots and normals: rank 0 adopts the builder's buffers, the other ranks' replicas are device-to-device copies of them.
    Three logical ranks on device 0 render the frame of the same scene committed through the host builder, bit for bit."""
    n = 36000
    rng = np.random.default_rng(11)
    d = scenes._base()
    d["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.7, 0.3, 0.16), param=0.0, texture=None))
    scenes._add_ref_spheres(d)
    c = rng.uniform(-2.0, 2.0, size=(n, 1, 3)).astype(np.float32) * np.array([1.0, 0.4, 1.0], np.float32) + np.array([0.0, 1.2, 0.0], np.float32)
    v = (c + rng.uniform(-0.05, 0.05, size=(n, 3, 3)).astype(np.float32)).reshape(3 * n, 3)
    d["meshes"].append(dict(vertices=np.ascontiguousarray(v), indices=np.arange(3 * n, dtype=np.uint32).reshape(n, 3), material=0))
    s = capi.Scene()
    for m in d["materials"]:
        s.add_material(m["type"], m["rgb"], m["param"], capi.NO_TEXTURE)
    for me in d["meshes"]:
        s.add_mesh(me["vertices"], me["indices"], me["material"])
    for sp in d["spheres"]:
        s.add_sphere(sp["center"], sp["radius"], sp["material"])
    s.commit_devices([0, 0, 0], capi.BUILD_AUTO)
    host = scenes.build(d, 0, 0)
    assert s.info().builder == 2 and host.info().builder == 0 and s.info().triangle_bytes == 48 * n
    W, H, spp = 160, 90, 3
    _, cam = scenes.make_camera(W)
    want, want_ids, st0 = host.render(cam, W, H, spp, want_ids=True)
    for mode in (capi.MULTI_TILE, capi.MULTI_SAMPLE):
        rgb, ids, st = s.render(cam, W, H, spp, want_ids=True, multi=mode)
        assert np.array_equal(ids, want_ids) and st.rays == st0.rays
        assert np.array_equal(rgb, want) if mode == capi.MULTI_TILE else np.allclose(rgb, want, rtol=3e-7, atol=1e-7)
    assert (want_ids >= 0).any() and ((want_ids >= 0) & (want_ids < n)).any()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,threads", [(1, "1"), (2, "0"), (3, "1"), (8, "0"), (8, "1")])
def test_multi_tile_mode_is_bitwise_the_single_device_frame(oracle, capi, scenes, monkeypatch, ranks, threads):
    monkeypatch.setenv("CRT_MULTI_THREADS", threads)             # 1: every rank's part of a frame is enqueued by a thread of its own
    desc = scenes.cats()
    s = _commit(scenes, capi, desc, [0] * ranks)
    W, H, spp = 203, 117, 4                                      # ragged
    params, cam = scenes.make_camera(W)
    single, ids1, st1 = s.render(cam, W, H, spp, want_ids=True)   # crt_render still works on a multi-device scene (devices[0])
    rgb, ids, st = s.render(cam, W, H, spp, want_ids=True, multi=capi.MULTI_TILE)
    assert np.array_equal(rgb, single) and np.array_equal(ids, ids1) and st.rays == st1.rays
    fb, rids, _, rst = oracle.RefScene(desc).render(oracle.make_camera(params, W), W, H, spp)
    assert np.array_equal(rgb, fb) and np.array_equal(ids, rids) and st.rays == rst.rays
    # sample chunks + device pointers, asynchronous, several frames in flight over the two buffer sets
    import torch
    outs = [torch.zeros(H * W * 3, dtype=torch.float32, device="cuda") for _ in range(5)]
    stream = torch.cuda.Stream()
    seeds = (3, 5, 7, 11, 1984)
    for seed, o in zip(seeds, outs):
        s.render(cam, W, H, spp, seed=seed, out_rgb=o.data_ptr(), want_stats=False, stream=stream.cuda_stream, sample_chunks=2,
                 flags=capi.OUT_DEVICE, multi=capi.MULTI_TILE)
    stream.synchronize()
    for seed, o in zip(seeds, outs):
        want = s.render(cam, W, H, spp, seed=seed, sample_chunks=2)[0]
        assert np.array_equal(o.cpu().numpy().reshape(want.shape), want), seed


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,threads", [(1, "0"), (2, "1"), (4, "0"), (4, "1")])
def test_multi_sample_mode_sums_rank_parts_in_rank_order(capi, scenes, monkeypatch, ranks, threads):
    import torch
    monkeypatch.setenv("CRT_MULTI_THREADS", threads)
    desc = scenes.cats()
    s = _commit(scenes, capi, desc, [0] * ranks)
    W, H, spp = 160, 90, 8
    _, cam = scenes.make_camera(W)
    single, ids1, st1 = s.render(cam, W, H, spp, want_ids=True, ids_sample=5)
    rgb, ids, st = s.render(cam, W, H, spp, want_ids=True, ids_sample=5, multi=capi.MULTI_SAMPLE)
    assert st.rays == st1.rays and np.array_equal(ids, ids1)
    # the copy transport adds the ranks' raw sums in rank order: reproduce exactly from per-rank OUT_ACCUM renders
    total = None
    for r in range(ranks):
        acc = s.render(cam, W, H, spp // ranks, sample_offset=r * (spp // ranks), flags=capi.OUT_ACCUM)[0]
        total = acc.copy() if total is None else (total + acc).astype(np.float32)
    v = (total / np.float32(spp)).astype(np.float32)
    want = np.where(v > 0, np.sqrt(v, dtype=np.float32), np.float32(0)).astype(np.float32)
    assert np.array_equal(rgb, want.reshape(rgb.shape))
    assert np.allclose(rgb, single, rtol=3e-7, atol=1e-7)        # float association only


@pytest.mark.gpu
def test_multi_over_rccl_single_device_communicator(capi, scenes, monkeypatch):
    """CRT_MULTI_TRANSPORT=rccl with one device: ncclCommInitAll, ncclAllGather / ncclReduce inside a group, then the
    epilogue -- the code path eight distinct devices take, with a world of one."""
    monkeypatch.setenv("CRT_MULTI_TRANSPORT", "rccl")
    desc = scenes.cats()
    s = _commit(scenes, capi, desc, [0])
    W, H, spp = 203, 117, 4
    _, cam = scenes.make_camera(W)
    single, ids1, st1 = s.render(cam, W, H, spp, want_ids=True)
    for mode in (capi.MULTI_TILE, capi.MULTI_SAMPLE):
        for _ in range(3):                                       # both buffer sets, and one of them twice
            rgb, ids, st = s.render(cam, W, H, spp, want_ids=True, multi=mode)
            assert np.array_equal(rgb, single) and np.array_equal(ids, ids1) and st.rays == st1.rays
    with pytest.raises(capi.CrtError):
        _commit(scenes, capi, desc, [0, 0])                      # a communicator refuses duplicate devices: reported, not hung


@pytest.mark.gpu
def test_reference_style_driver_with_gpus_flag(oracle, scenes, tmp_path):
    """examples/crt_demo --gpus N: Renderer::devices drives crt_scene_commit_devices / crt_render_multi.  One logical rank on
    device 0 must give the single-device frame bit for bit; three logical ranks on device 0 as well (tile mode)."""
    demo = os.path.join(helpers.REPO, "examples", "crt_demo")
    W, H, spp = 200, 120, 3
    params, _ = scenes.make_camera(W)
    fb, _, _, _ = oracle.RefScene(scenes.cat()).render(oracle.make_camera(params, W), W, H, spp)
    want = np.clip(fb * np.float32(255.0), 0, 255).astype(np.uint8)
    header = ("P6\n%d %d\n255\n" % (W, H)).encode()
    for gpus in ("1", "0,0,0"):
        r = subprocess.run([demo, os.path.join(helpers.REPO, "models", "cat.obj"), str(W), str(H), str(spp), "--gpus", gpus], cwd=tmp_path,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        raw = open(tmp_path / "output.ppm", "rb").read()
        img = np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(H, W, 3)
        assert np.array_equal(img, want), gpus


@pytest.mark.gpu
def test_multi_and_noalloc_argument_errors(capi, scenes):
    """Error behaviour of the round-2 entry points: every misuse returns a code and a message, nothing is rendered."""
    import torch
    desc = scenes.cat()
    single = scenes.build(desc, 0)
    _, cam = scenes.make_camera(64)
    out = torch.zeros(36 * 64 * 3, dtype=torch.float32, device="cuda")
    with pytest.raises(capi.CrtError) as e:                       # crt_render_multi needs crt_scene_commit_devices
        single.render(cam, 64, 36, 2, multi=capi.MULTI_TILE)
    assert e.value.code == capi.CRT_ERR_INVALID and "crt_scene_commit_devices" in str(e.value)
    s = _commit(scenes, capi, desc, [0, 0, 0])
    for kw, needle in [(dict(tile_rank=1, tile_ranks=3), "partitions the frame itself"), (dict(flags=capi.OUT_ACCUM), "partitions the frame itself"),
                       (dict(multi=7), "unknown mode")]:
        mode = kw.pop("multi", capi.MULTI_TILE)
        with pytest.raises(capi.CrtError) as e:
            s.render(cam, 64, 36, 2, multi=mode, **kw)
        assert e.value.code == capi.CRT_ERR_INVALID and needle in str(e.value), (kw, str(e.value))
    with pytest.raises(capi.CrtError) as e:                       # 4 samples do not split over 3 ranks
        s.render(cam, 64, 36, 4, multi=capi.MULTI_SAMPLE)
    assert "multiple of the number of ranks" in str(e.value)
    with pytest.raises(capi.CrtError) as e:
        s.commit_devices([0, 99])
    assert e.value.code == capi.CRT_ERR_NO_DEVICE
    with pytest.raises(capi.CrtError) as e:
        s.commit_devices([])
    assert e.value.code == capi.CRT_ERR_INVALID
    # after a failed commit the handle is uncommitted, and says so
    with pytest.raises(capi.CrtError):
        s.render(cam, 64, 36, 1)
    # crt_render_noalloc: same parameter checks as crt_render
    with pytest.raises(capi.CrtError) as e:
        capi.render_noalloc(single, cam, 64, 36, 3, out.data_ptr(), sample_chunks=2)
    assert e.value.code == capi.CRT_ERR_INVALID and "sample_chunks" in str(e.value)
    with pytest.raises(capi.CrtError) as e:
        capi.render_noalloc(single, cam, 0, 36, 1, out.data_ptr())
    assert e.value.code == capi.CRT_ERR_INVALID
    # the tuning knobs are read at commit: without a reload a changed environment changes nothing
    os.environ["CRT_KERNEL"] = "1"
    try:
        was = capi.RELOAD_TUNING_EACH_RENDER
        capi.RELOAD_TUNING_EACH_RENDER = False
        a = single.render(cam, 64, 36, 2)[0]
        single.reload_tuning()                                    # now the pool kernel renders: same image, bit for bit
        b = single.render(cam, 64, 36, 2)[0]
        assert np.array_equal(a, b)
    finally:
        capi.RELOAD_TUNING_EACH_RENDER = was
        del os.environ["CRT_KERNEL"]
        single.reload_tuning()
