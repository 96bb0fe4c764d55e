"""world_size-2 (and 3) gloo tests of the multi-rank frame assembly (cudaraytracer_amd/dist.py) on the CPU.

No GPU here, so each rank produces its share with the test-only host single-stepper (the product's per-lane code
compiled for the host) and the de-interleave / finalize epilogues are numpy stand-ins supplied by the test; what is
under test is dist.py's partitioning, buffer sizing and collectives, against the oracle's full frame.
"""
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

    if mode == "tile":
        f = TileFrame(W, H, world, rank, torch.device("cpu"), untile_fn=untile)
        part, _, st = hs.render(s, cam, W, H, SPP, tile_rank=rank, tile_ranks=world)
        f.local.copy_(torch.from_numpy(helpers.to_compact(part, rank, world)))
        frame = f.gather()
    else:
        f = SampleFrame(W, H, SPP, world, rank, torch.device("cpu"), finalize_fn=finalize)
        part, _, st = hs.render(s, cam, W, H, f.spp_local, sample_offset=rank * f.spp_local, flags=1)
        f.accum.copy_(torch.from_numpy(part.reshape(-1)))
        frame = f.reduce()
    rays = torch.tensor([st["rays"]], dtype=torch.int64)
    dist.all_reduce(rays)
    if mode == "tile" or rank == 0:
        np.save(os.path.join(outdir, "frame_%s_%d.npy" % (mode, rank)), frame.numpy().reshape(H, W, 3))
    if rank == 0:
        np.save(os.path.join(outdir, "rays_%s.npy" % mode), rays.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("tile", 2), ("tile", 3), ("sample", 2)])
def test_frame_assembly_over_gloo(oracle, scenes, hostsim, tmp_path, mode, world):
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    desc = scenes.cats()
    params, _ = scenes.make_camera(W)
    fb, _, _, st = oracle.RefScene(desc).render(oracle.make_camera(params, W), W, H, SPP)
    assert int(np.load(tmp_path / ("rays_%s.npy" % mode))[0]) == st.rays
    if mode == "tile":
        for r in range(world):   # every rank holds the whole frame, bitwise equal to the single-process oracle frame
            assert np.array_equal(np.load(tmp_path / ("frame_tile_%d.npy" % r)), fb)
    else:                        # sample split: float sum order differs -> within 2 ulp of the gamma-encoded value
        got = np.load(tmp_path / "frame_sample_0.npy")
        assert np.allclose(got, fb, rtol=3e-7, atol=1e-7)
