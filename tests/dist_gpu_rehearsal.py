"""Helper of tests/test_dist.py::test_ranks_sharing_one_gpu_over_gloo (run under torch.distributed.run): N ranks on device 0 over
gloo, the real library and the real dist.py.  Every rank renders its tiles / its samples; the assembled frame must be the frame
one rank renders alone (tile mode: bit for bit; sample mode: to float association), for a few frames in a row through the
pipelined slots.  Prints 'REHEARSAL OK <mode>' on rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracer_amd import capi, scenes  # noqa: E402
from cudaraytracer_amd.dist import SampleFrame, TileFrame  # noqa: E402


def main():
    mode = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    W, H, spp = 203, 117, 4 * world   # ragged frame
    s = scenes.build(scenes.cats(), 0)
    _, cam = scenes.make_camera(W)
    seeds = (7, 11, 1984)
    if mode == "tile":
        f = TileFrame(W, H, world, rank, dev, sample_chunks=spp, pipelined=2)
        got = []
        for seed in seeds:
            f.render_local(s, cam, spp, 10, seed)
            f.gather_async()
            f.drain()
            torch.cuda.synchronize()
            got.append(f.frame.cpu().numpy().copy())
        for seed, g in zip(seeds, got):
            want = s.render(cam, W, H, spp, seed=seed, sample_chunks=spp)[0]
            assert np.array_equal(g.reshape(want.shape), want), (rank, seed)
    else:
        f = SampleFrame(W, H, spp, world, rank, dev, sample_chunks=spp, pipelined=2)
        rays = 0
        for seed in seeds:
            st = f.render_local(s, cam, 10, seed, want_stats=True)
            rays = st.rays
            f.reduce_async()
            f.drain()
            torch.cuda.synchronize()
            if rank == 0:
                want = s.render(cam, W, H, spp, seed=seed)[0]
                assert np.allclose(f.frame.cpu().numpy().reshape(want.shape), want, rtol=3e-7, atol=1e-7), seed
        t = torch.tensor([rays], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            assert int(t[0]) == s.render(cam, W, H, spp, seed=seeds[-1])[2].rays
    dist.barrier()
    if rank == 0:
        print("REHEARSAL OK", mode, world, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
