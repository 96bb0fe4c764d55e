"""The three tree builders (host SAH, GPU LBVH, GPU SAH) on the bench workloads: build time, tree statistics, surface-area
cost, and the frame time of the same frame rendered from each tree, interleaved over `reps` rounds on one box.  The frames must
be identical (ids, image, ray count): boxes only prune.
usage: python tools/builder_compare.py [cats|cat_grid|cat ...] [--scale S] [--reps R]"""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from cudaraytracer_amd import capi, scenes

names = [a for a in sys.argv[1:] if a.isidentifier()] or ['cats', 'cat_grid']
scale = int(sys.argv[sys.argv.index('--scale') + 1]) if '--scale' in sys.argv else 1
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 5
BUILDERS = [('host_sah', 0), ('gpu_lbvh', capi.BUILD_GPU_LBVH), ('gpu_sah', capi.BUILD_GPU_SAH)]
for name in names:
    W, H, spp, cam_kw = (3840 // scale, 2160 // scale, 16, scenes.GRID_CAMERA) if name == 'cat_grid' else (1920 // scale, 1080 // scale, 64, {})
    desc = getattr(scenes, name)()
    _, cam = scenes.make_camera(W, **cam_kw)
    built = {}
    for label, flags in BUILDERS:
        t0 = time.perf_counter()
        s = scenes.build(desc, 0, flags)
        wall = (time.perf_counter() - t0) * 1e3
        if flags:  # second build: the first one pays for the module load and the allocator's first touches
            t0 = time.perf_counter()
            s = scenes.build(desc, 0, flags)
            wall = (time.perf_counter() - t0) * 1e3
        built[label] = (s, wall)
    out = torch.empty(W * H * 3, dtype=torch.float32, device='cuda')
    ids = torch.empty(W * H, dtype=torch.int32, device='cuda')
    frames, times = {}, {k: [] for k, _ in BUILDERS}
    for label, _ in BUILDERS:
        s = built[label][0]
        st = s.render(cam, W, H, spp, out_rgb=out.data_ptr(), out_ids=ids.data_ptr(), sample_chunks=spp, flags=capi.OUT_DEVICE)[2]
        torch.cuda.synchronize()
        frames[label] = (out.cpu().numpy().copy(), ids.cpu().numpy().copy(), st.rays)
    for _ in range(reps):
        for label, _ in BUILDERS:
            s = built[label][0]
            times[label].append(s.render(cam, W, H, spp, out_rgb=out.data_ptr(), sample_chunks=spp, flags=capi.OUT_DEVICE)[2].kernel_ms)
    ref = frames['host_sah']
    for label, _ in BUILDERS:
        s, wall = built[label]
        i = s.info()
        f = frames[label]
        same = bool(np.array_equal(f[0], ref[0]) and np.array_equal(f[1], ref[1]) and f[2] == ref[2])
        st = s.render(cam, W, H, spp, out_rgb=out.data_ptr(), flags=capi.COUNT_WORK | capi.OUT_DEVICE, sample_chunks=spp)[2]
        print(json.dumps(dict(scene=name, frame="%dx%dx%d" % (W, H, spp), builder=label, builder_id=i.builder, build_ms=round(i.build_ms, 2),
                              commit_wall_ms=round(wall, 1), nodes=i.num_nodes, depth=i.bvh_depth, sah_cost=round(i.sah_cost, 3),
                              frame_ms_min=round(min(times[label]), 3), frame_ms_median=round(float(np.median(times[label])), 3),
                              vs_host=round(float(np.median(times[label]) / np.median(times['host_sah'])), 4),
                              nodes_per_ray=round(st.nodes / st.rays, 3), tris_per_ray=round(st.tri_tests / st.rays, 3), same_frame=same)), flush=True)
