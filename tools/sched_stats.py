import sys; sys.path.insert(0,'.')
import torch
from cudaraytracer_amd import capi, scenes
s = scenes.build(scenes.cats(), 0)
W,H,spp=1920,1080,64
_, cam = scenes.make_camera(W)
out = torch.empty(W*H*3, dtype=torch.float32, device='cuda')
for it in range(2):
    _,_,st = s.render(cam, W,H,spp, out_rgb=out.data_ptr(), flags=capi.COUNT_WORK)
w = list(st.sched)
print('rays',st.rays,'nodes',st.nodes,'tris',st.tri_tests,'shaded',st.shaded_hits,'ms',st.kernel_ms)
nw=32400
print('A runs/wave %.1f ready lanes/run %.1f (HIT lanes %.1f) ideal runs/wave %.1f' % (w[0]/nw, w[1]/max(w[0],1), w[2]/max(w[0],1), w[1]/64/nw))
print('INNER steps/wave %.1f lanes/step %.1f ideal %.1f' % (w[4]/nw, w[5]/max(w[4],1), w[5]/64/nw))
print('LEAF  steps/wave %.1f lanes/step %.1f ideal %.1f' % (w[6]/nw, w[7]/max(w[6],1), w[7]/64/nw))
