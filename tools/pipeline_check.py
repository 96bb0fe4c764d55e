"""Does the collective path stay off the critical path?  A frame of 1/8 of the headline's pixels (what one of 8 ranks
renders), K steps: two frames in flight without a collective, then with the all-gather + de-interleave of a 1-rank RCCL
group behind every frame, then one frame at a time.  usage: pipeline_check.py [GPU_MAX_HW_QUEUES]"""
import os, socket, sys, time
if len(sys.argv) > 1:
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]   # as bench.py does: set in the process, before HIP starts
sys.path.insert(0, '.')
import torch
import torch.distributed as dist
from cudaraytracer_amd import scenes
from cudaraytracer_amd.dist import TileFrame
sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
s = scenes.build(scenes.cats(), 0)
W, H, spp, K = 680, 382, 64, 40
_, cam = scenes.make_camera(W)
CASES = (("two in flight, no collective", dict(pipelined=True)), ("three in flight, all-gather + untile", dict(pipelined=3, force_collective=True)), ("two in flight, all-gather + untile", dict(pipelined=True, force_collective=True)),
                 ("one at a time, all-gather + untile", dict(pipelined=False, force_collective=True)), ("one at a time, no collective", dict(pipelined=False)))
for name, kw in CASES + CASES[::-1]:   # every case twice, in both orders: the first case also pays RCCL's first-use set-up
    f = TileFrame(W, H, 1, 0, dev, sample_chunks=64, **kw)
    for _ in range(4):
        f.render_local(s, cam, spp, 10, 1984); f.gather_async()
    f.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        f.render_local(s, cam, spp, 10, 1984); f.gather_async()
    f.drain(); torch.cuda.synchronize()
    print("%-38s %.3f ms per step (GPU_MAX_HW_QUEUES=%s)" % (name, (time.perf_counter() - t0) / K * 1e3, os.environ.get("GPU_MAX_HW_QUEUES", "default")))
dist.destroy_process_group()
