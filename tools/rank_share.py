"""What one rank of an N-way tile partition costs on this GPU: the headline frame's tiles k = rank (mod N), compact output,
K frames back to back, against 1/N of the whole frame.  Each case is timed with n frames in flight on n streams, n from INFLIGHT (default 1,2).
usage: [INFLIGHT=1,2,3] rank_share.py [N] [chunks]"""
import sys, time
sys.path.insert(0, '.')
import torch
from cudaraytracer_amd import capi, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H, spp, K = 1920, 1080, 64, 20
s = scenes.build(scenes.cats(), 0)
_, cam = scenes.make_camera(W)
full = torch.empty(W * H * 3, dtype=torch.float32, device='cuda')
part = torch.empty(capi.compact_floats(W, H, N), dtype=torch.float32, device='cuda')
streams = [torch.cuda.Stream() for _ in range(4)]
full2 = [torch.empty_like(full) for _ in range(4)]
part2 = [torch.empty_like(part) for _ in range(4)]
def timed(fn, inflight):
    for k in range(4): fn(k % inflight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): fn(k % inflight)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
import os
for inflight in [int(v) for v in os.environ.get("INFLIGHT", "1,2").split(",")]:
    t_full = timed(lambda b: s.render(cam, W, H, spp, out_rgb=full2[b].data_ptr(), sample_chunks=chunks, want_stats=False, stream=streams[b].cuda_stream), inflight)
    ts = []
    for r in range(N):
        ts.append(timed(lambda b: s.render(cam, W, H, spp, out_rgb=part2[b].data_ptr(), sample_chunks=chunks, want_stats=False, stream=streams[b].cuda_stream,
                                           tile_rank=r, tile_ranks=N, flags=capi.OUT_COMPACT), inflight))
    print("N=%d chunks=%d frames in flight %d: whole frame %.3f ms; ranks %.3f .. %.3f ms, efficiency with the slowest rank %.3f"
          % (N, chunks, inflight, t_full, min(ts), max(ts), t_full / N / max(ts)))
