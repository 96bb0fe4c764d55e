"""crt_render_noalloc under HIP stream capture (torch.cuda.graph): a frame large enough for the two-launch form (helper stream,
fork / join events) is captured once and replayed; the replayed frame must equal crt_render's."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudaraytracer_amd import capi, scenes

s = scenes.build(scenes.cats(), 0)
for W, H, SPP, CH, nc in ((1536, 864, 8, 2, "4"), (480, 270, 4, 4, "1")):
    os.environ["CRT_NODE_CACHE"] = nc
    s.reload_tuning()
    _, cam = scenes.make_camera(W)
    want = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
    s.render(cam, W, H, SPP, out_rgb=want.data_ptr(), sample_chunks=CH, flags=capi.OUT_DEVICE)
    got = torch.zeros_like(want)
    stream = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        capi.render_noalloc(s, cam, W, H, SPP, got.data_ptr(), sample_chunks=CH, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("captured %dx%d; output untouched by the capture itself: %s" % (W, H, bool((got == 0).all().item())), flush=True)
    for k in range(2):
        got.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        print("replay %d equals crt_render: %s" % (k, torch.equal(got, want)), flush=True)
