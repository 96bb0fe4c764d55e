"""node_cache = 6 (one 16-wave + one 4-wave workgroup per CU, two launches on one work queue) against the default: same frame?"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from cudaraytracer_amd import capi, scenes
capi.RELOAD_TUNING_EACH_RENDER = True
s = scenes.build(scenes.cats(), 0)
for W, H, spp in ((200, 113, 8), (1920, 1080, 64)):
    _, cam = scenes.make_camera(W)
    os.environ.pop("CRT_NODE_CACHE", None)
    a = s.render(cam, W, H, spp, want_ids=True, sample_chunks=spp)
    os.environ["CRT_NODE_CACHE"] = "6"
    b = s.render(cam, W, H, spp, want_ids=True, sample_chunks=spp)
    print(W, H, spp, "equal:", np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1]), a[2].rays == b[2].rays, "kernel ms", round(a[2].kernel_ms, 3), round(b[2].kernel_ms, 3), flush=True)
