"""Two-child against four-child nodes (CRT_BVH_WIDE) on one bench workload: plain kernel time and the instrumented launch's
work counters.  usage: python tools/wide_stats.py [cat_grid|cats] [scale]   (scale divides the frame's width and height)"""
import json, os, sys
sys.path.insert(0, '.')
import torch
from cudaraytracer_amd import capi, scenes
name = sys.argv[1] if len(sys.argv) > 1 else 'cat_grid'
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H, spp, cam_kw = (3840 // scale, 2160 // scale, 16, scenes.GRID_CAMERA) if name == 'cat_grid' else (1920 // scale, 1080 // scale, 64, {})
for wide in ("0", "1"):
    os.environ["CRT_BVH_WIDE"] = wide
    s = scenes.build(getattr(scenes, name)(), 0)
    info = s.info()
    _, cam = scenes.make_camera(W, **cam_kw)
    out = torch.empty(W * H * 3, dtype=torch.float32, device='cuda')
    chunks = spp
    s.render(cam, W, H, spp, out_rgb=out.data_ptr(), sample_chunks=chunks, flags=capi.OUT_DEVICE)
    ms = min(s.render(cam, W, H, spp, out_rgb=out.data_ptr(), sample_chunks=chunks, flags=capi.OUT_DEVICE)[2].kernel_ms for _ in range(3))
    st = s.render(cam, W, H, spp, out_rgb=out.data_ptr(), flags=capi.COUNT_WORK | capi.OUT_DEVICE, sample_chunks=chunks)[2]
    w = list(st.sched)
    print(json.dumps(dict(width=info.node_width, depth=info.bvh_depth, nodes_total=info.num_nodes, ms=round(ms, 3), rays=st.rays,
                          nodes_per_ray=round(st.nodes / st.rays, 2), tris_per_ray=round(st.tri_tests / st.rays, 2),
                          a_runs=w[0], a_lanes_mean=round(w[1] / max(w[0], 1), 1), inner_steps=w[4], inner_lanes_mean=round(w[5] / max(w[4], 1), 1),
                          leaf_steps=w[6], leaf_lanes_mean=round(w[7] / max(w[6], 1), 1), count_work_ms=round(st.kernel_ms, 2))), flush=True)
