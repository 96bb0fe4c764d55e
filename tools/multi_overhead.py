"""What the one-process multi-GPU path (crt_scene_commit_devices / crt_render_multi) costs on ONE GPU: the headline frame
rendered by n logical ranks that all map to device 0 (peer copies instead of RCCL), asynchronously, against crt_render with
two frames in flight.  The n partial renders share the GPU, so ms/frame shows the overhead of partitioning, exchange and
epilogue, not a speed-up."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracer_amd import capi, scenes
from tests.test_multi import _commit

W, H, spp, frames = 1920, 1080, 64, 20
desc = scenes.cats()
_, cam = scenes.make_camera(W)
out = [torch.empty(W * H * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
res = {}
for ranks in (1, 2, 4, 8):
    for mode, name in ((capi.MULTI_TILE, "tile"), (capi.MULTI_SAMPLE, "sample")):
        s = _commit(scenes, capi, desc, [0] * ranks)
        stream = torch.cuda.Stream()
        enqueue = []
        def go(n):
            for k in range(n):
                t = time.perf_counter()
                s.render(cam, W, H, spp, out_rgb=out[k % 2].data_ptr(), want_stats=False, stream=stream.cuda_stream, sample_chunks=32,
                         flags=capi.OUT_DEVICE, multi=mode)
                enqueue.append(time.perf_counter() - t)
            stream.synchronize()
        go(3)
        del enqueue[:]
        t0 = time.perf_counter(); go(frames); dt = time.perf_counter() - t0
        res["%s x%d" % (name, ranks)] = round(dt / frames * 1e3, 3)
        # the host's share: the time the asynchronous call itself takes (it returns when everything is enqueued; once the queues are
        # full it also waits for room, so the median and the first calls after a sync tell more than the mean)
        enqueue.sort()
        res["%s x%d enqueue ms (min / median)" % (name, ranks)] = [round(enqueue[0] * 1e3, 3), round(enqueue[len(enqueue) // 2] * 1e3, 3)]
        s.close()
s = scenes.build(desc, 0)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def go1(n):
    for k in range(n):
        s.render(cam, W, H, spp, out_rgb=out[k % 2].data_ptr(), want_stats=False, stream=streams[k % 2].cuda_stream, sample_chunks=32, flags=capi.OUT_DEVICE)
    for st in streams: st.synchronize()
go1(3)
t0 = time.perf_counter(); go1(frames); dt = time.perf_counter() - t0
res["crt_render, two streams"] = round(dt / frames * 1e3, 3)
print(json.dumps(res))
