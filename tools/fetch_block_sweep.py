"""Work units per trip to a block counter (CRT_FETCH_BLOCK) against the share of the headline frame a rank renders:
rank 0 of N = 1, 2, 4, 8 tile ranks, frames back to back with bench.py's number of frames in flight.  usage: fetch_block_sweep.py [chunks]"""
import os, sys, time
sys.path.insert(0, '.')
import torch
from cudaraytracer_amd import capi, scenes
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H, spp, K = 1920, 1080, 64, 20
s = scenes.build(scenes.cats(), 0)
_, cam = scenes.make_camera(W)
streams = [torch.cuda.Stream() for _ in range(3)]
bufs = [torch.empty(W * H * 3, dtype=torch.float32, device='cuda') for _ in range(3)]
def timed(N, inflight):
    def fn(b):
        s.render(cam, W, H, spp, out_rgb=bufs[b].data_ptr(), sample_chunks=chunks, want_stats=False, stream=streams[b].cuda_stream,
                 tile_rank=0, tile_ranks=N, flags=capi.OUT_COMPACT | capi.OUT_DEVICE)
    for k in range(4): fn(k % inflight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): fn(k % inflight)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
blocks = [int(v) for v in os.environ.get("BLOCKS", "128,256,512,1024").split(",")]
for rep in range(2):
    for N in (1, 2, 4, 8):
        row = []
        for fb in blocks:
            os.environ["CRT_FETCH_BLOCK"] = str(fb)
            s.reload_tuning()
            row.append("%d: %.3f" % (fb, timed(N, 2)))
        print("ranks %d, chunks %d, ms per rank-frame by block size  " % (N, chunks) + "   ".join(row), flush=True)
