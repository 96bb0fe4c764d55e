"""Writes the regression fixtures under tests/golden/ from the oracle (oracle/crt_ref.c).

The reference repository holds no golden vectors and cannot be built here, so these fixtures are
outputs of the CPU restatement, committed so that the oracle itself cannot drift silently:
  cat_400x400_primary_ids.npz : BASELINE config 1 (cat.obj, 400x400, 16 spp, depth 10, seed 1984):
      ids            int16 [400,400]  primary-hit primitive id of sample 0 (-1 miss; 732 = ground, 733 = light)
      rays_16spp     total path segments
      tile_means_16spp float32 [50,50,3] means of the gamma image over 8x8 tiles
      sha256_16spp   digest of the float32 image bytes
  cats_1080p_crop_ids.npz : BASELINE config 3 scene at 1920x1080, sample 0, window x 832..1088, y 476..604
Run:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cudaraytracer_amd import scenes  # noqa: E402  (scene descriptions + camera parameters only)
from oracle import binding as ob      # noqa: E402


def main():
    rs = ob.RefScene(scenes.cat())
    params, _ = scenes.make_camera(400)
    cam = ob.make_camera(params, 400)
    fb, ids, _, st = rs.render(cam, 400, 400, 16)
    np.savez_compressed(os.path.join(HERE, "cat_400x400_primary_ids.npz"), ids=ids.astype(np.int16), rays_16spp=np.uint64(st.rays),
                        tile_means_16spp=fb.reshape(50, 8, 50, 8, 3).mean(axis=(1, 3)).astype(np.float32),
                        sha256_16spp=hashlib.sha256(fb.tobytes()).hexdigest())
    rs = ob.RefScene(scenes.cats())
    params, _ = scenes.make_camera(1920)
    cam = ob.make_camera(params, 1920)
    win = (832, 476, 1088, 604)
    fb, ids, _, st = rs.render(cam, 1920, 1080, 1, window=win)
    np.savez_compressed(os.path.join(HERE, "cats_1080p_crop_ids.npz"), ids=ids[win[1]:win[3], win[0]:win[2]].astype(np.int16),
                        window=np.array(win), rays=np.uint64(st.rays),
                        sha256=hashlib.sha256(fb[win[1]:win[3], win[0]:win[2]].tobytes()).hexdigest())
    print("wrote fixtures; cat ids hit fraction", float((ids >= 0).mean()))


if __name__ == "__main__":
    main()
