import sys, time, os
sys.path.insert(0, '.')
import torch
from cudaraytracer_amd import capi, scenes
W, H, spp = 1000, 1000, 10
chunks = int(sys.argv[1])
s = scenes.build(scenes.cat(), 0)
_, cam = scenes.make_camera(W)
out = torch.empty(W * H * 3, dtype=torch.float32, device='cuda')
torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    st = s.render(cam, W, H, spp, out_rgb=out.data_ptr(), sample_chunks=chunks, flags=capi.OUT_DEVICE)[2]
    t1 = time.perf_counter()
    print("chunks %d frame %d: %.3f ms wall, kernel %.3f ms" % (chunks, k, (t1 - t0) * 1e3, st.kernel_ms))
