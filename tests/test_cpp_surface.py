"""The drop-in C++ surface (cudaraytracer_amd/cpp/*.h: Scene / Renderer / Camera / LoadModel / Model with the reference's
signatures) driven by examples/main.cpp, which follows the reference's main() (kernel.cu:44-103)."""
import os
import subprocess

import numpy as np
import pytest

from tests import helpers

DEMO = os.path.join(helpers.REPO, "examples", "crt_demo")
RPATH = os.path.join(helpers.REPO, "cudaraytracer_amd")


def _build(tmp_path):
    exe = str(tmp_path / "crt_demo")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(helpers.REPO, "examples", "main.cpp"),
                           "-L" + RPATH, "-lcrt_hip", "-Wl,-rpath," + RPATH, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe])
    return exe


def test_surface_compiles_and_fails_loudly_without_gpu(capi, tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, os.path.join(helpers.REPO, "models", "cat.obj"), "32", "32", "1"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 99 and "CRT error" in r.stderr        # checkCrtErrors keeps the reference's exit(99) (utility.h:31-40)
    assert "Scene Created with 734 objects" in r.stdout          # scene.h:175


@pytest.mark.gpu
@pytest.mark.parametrize("spp", [1, 3, 4, 10])   # Renderer's default is one sample per work unit for any count (10: the reference's own, kernel.cu:49)
def test_reference_style_driver_matches_oracle(oracle, scenes, tmp_path, spp):
    assert os.path.exists(DEMO), "examples/crt_demo not built (python __graft_entry__.py)"
    W, H = 200, 120
    r = subprocess.run([DEMO, os.path.join(helpers.REPO, "models", "cat.obj"), str(W), str(H), str(spp)], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(tmp_path / "output.ppm", "rb").read()
    header = ("P6\n%d %d\n255\n" % (W, H)).encode()
    assert raw.startswith(header)
    img = np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(H, W, 3)
    params, _ = scenes.make_camera(W)
    fb, _, _, _ = oracle.RefScene(scenes.cat()).render(oracle.make_camera(params, W), W, H, spp)
    want = np.clip(fb * np.float32(255.0), 0, 255).astype(np.uint8)   # utility.h:133-137: clamp, truncate
    assert np.array_equal(img, want)
    from tests.test_output import decode_png
    assert np.array_equal(decode_png(str(tmp_path / "output.png")), want)   # export_framebuffer_to_png (utility.h:122-145)


@pytest.mark.gpu
@pytest.mark.parametrize("builder", ["gpusah", "lbvh", "host"])
def test_reference_style_driver_with_each_builder(tmp_path, builder):
    """CRT_BUILDER through the C++ surface (Scene::launch_triangles_scene -> crt_scene_commit): whoever builds the tree, the
    driver writes the same file, byte for byte."""
    model = os.path.join(helpers.REPO, "models", "cats.obj")
    outs = []
    for env_builder in (None, builder):
        d = tmp_path / (env_builder or "default")
        d.mkdir()
        env = dict(os.environ)
        env.pop("CRT_BUILDER", None)
        if env_builder:
            env["CRT_BUILDER"] = env_builder
        r = subprocess.run([DEMO, model, "192", "108", "3"], cwd=d, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        outs.append(open(d / "output.ppm", "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 192 * 108 * 3


@pytest.mark.gpu
def test_reference_style_driver_sphere_scene(tmp_path):
    """launch_random_scene_kernel (scene.h:197-235) through the C++ surface: 488 spheres, PNG out.  The C++ generator's
    numbers are its own, so the check is on the frame's content, not against the oracle (tests/test_gpu.py does that for
    the python-described scene)."""
    from tests.test_output import decode_png
    r = subprocess.run([DEMO, "spheres", "320", "180", "8"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Scene Created with 488 objects" in r.stdout
    img = decode_png(str(tmp_path / "output.png"))
    assert img.shape == (180, 320, 3)
    sky = img[:20].reshape(-1, 3).mean(axis=0)
    assert sky[2] > sky[0] > 150 and img[100:].std() > 20      # blue-ish sky on top, structure below


@pytest.mark.gpu
def test_reference_style_driver_large_frame_many_samples(tmp_path):
    """Renderer's default (one sample per work unit) at a size whose chunk slabs would be 25 GB (3840 x 2160 x 256 x 12 B): the
    library must keep its scratch bounded by itself (per-wave rings beyond the slab budget) instead of failing or allocating it."""
    r = subprocess.run([DEMO, os.path.join(helpers.REPO, "models", "cat.obj"), "3840", "2160", "256"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from tests.test_output import decode_png
    img = decode_png(str(tmp_path / "output.png"))
    assert img.shape == (2160, 3840, 3) and img.std() > 10
