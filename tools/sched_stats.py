"""Schedule statistics of the render kernel on the bench workload: one plain launch (for rocprofv3 --pmc) and one
instrumented launch (CRT_COUNT_WORK) whose wave-level schedule counters are printed as JSON."""
import json, sys
sys.path.insert(0, '.')
import torch
from cudaraytracer_amd import capi, scenes
name = sys.argv[1] if len(sys.argv) > 1 else 'cats'
import os
chunks = int(os.environ.get('CHUNKS', '0'))
W, H, spp = 1920, 1080, 64
s = scenes.build(getattr(scenes, name)(), 0)
_, cam = scenes.make_camera(W)
out = torch.empty(W * H * 3, dtype=torch.float32, device='cuda')
s.render(cam, W, H, spp, out_rgb=out.data_ptr(), sample_chunks=chunks)
_, _, st0 = s.render(cam, W, H, spp, out_rgb=out.data_ptr(), sample_chunks=chunks)
_, _, st = s.render(cam, W, H, spp, out_rgb=out.data_ptr(), flags=capi.COUNT_WORK, sample_chunks=chunks)
w = list(st.sched)
print(json.dumps(dict(ms=st0.kernel_ms, rays=st.rays, nodes=st.nodes, tris=st.tri_tests, shaded=st.shaded_hits, a_runs=w[0], a_lanes=w[1], 
                      starved_lanes=w[3], inner_steps=w[4], inner_lanes=w[5], leaf_steps=w[6], leaf_lanes=w[7],
                      waves=w[8], wave_ms_mean=w[9] / max(w[8], 1) * 1e-5, wave_ms_max=w[2] * 1e-5, count_work_ms=st.kernel_ms)))
