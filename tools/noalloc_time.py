"""crt_render_noalloc against crt_render on a frame that fills the GPU and fits the 32 MiB slab limit (cats, 1536 x 864,
64 spp as 2 sample chunks = 2.6 M work units): ms per frame of each entry point, frames back to back on one stream."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudaraytracer_amd import capi, scenes

s = scenes.build(scenes.cats(), 0)
W, H, SPP, CH = 1536, 864, 64, 2
_, cam = scenes.make_camera(W)
rgb = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream()


def run(kind, n):
    for _ in range(n):
        if kind == "render":
            s.render(cam, W, H, SPP, out_rgb=rgb.data_ptr(), sample_chunks=CH, flags=capi.OUT_DEVICE, stream=stream.cuda_stream, want_stats=False)
        else:
            capi.render_noalloc(s, cam, W, H, SPP, rgb.data_ptr(), sample_chunks=CH, stream=stream.cuda_stream)
    stream.synchronize()


for kind in ("render", "noalloc", "render", "noalloc"):
    run(kind, 2)
    t = time.perf_counter()
    run(kind, 10)
    print(kind, "%.3f ms per frame" % ((time.perf_counter() - t) * 100.0), flush=True)
