import sys; sys.path.insert(0, '.')
import numpy as np
from cudaraytracer_amd import capi, scenes
s = scenes.build(scenes.cats(), 0)
for W, H, spp in ((200, 113, 64), (333, 187, 16), (1920, 1080, 64)):
    _, cam = scenes.make_camera(W)
    a = s.render(cam, W, H, spp, sample_chunks=1)[0]
    b = s.render(cam, W, H, spp, sample_chunks=spp)[0]
    c = s.render(cam, W, H, spp, sample_chunks=spp // 2)[0]
    print(W, H, spp, "chunks=spp equal to whole pixels:", np.array_equal(a, b), " chunks=spp/2 equal:", np.array_equal(a, c), flush=True)
