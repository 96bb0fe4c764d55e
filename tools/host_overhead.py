"""Host cost of one pipelined step (render_local + gather_async over a 1-rank RCCL group) on a frame so small that the GPU
is never the limit: what the enqueue path costs per frame, to be compared with 1/N of a frame at N ranks."""
import os, socket, sys, time
sys.path.insert(0, '.')
import torch
import torch.distributed as dist
from cudaraytracer_amd import scenes
from cudaraytracer_amd.dist import TileFrame
sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
s = scenes.build(scenes.cats(), 0)
W, H = 64, 64
_, cam = scenes.make_camera(W)
for coll in (False, True):
    f = TileFrame(W, H, 1, 0, dev, sample_chunks=4, pipelined=True, force_collective=coll)
    for _ in range(5):
        f.render_local(s, cam, 4, 10, 1984); f.gather_async()
    f.drain(); torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        f.render_local(s, cam, 4, 10, 1984); f.gather_async()
    f.drain(); torch.cuda.synchronize()
    print("collective=%s: %.1f us per step" % (coll, (time.perf_counter() - t0) / K * 1e6))
dist.destroy_process_group()
