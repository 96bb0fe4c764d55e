import sys, time, os
sys.path.insert(0, '.')
os.environ["CRT_PRINT_COMMIT"] = "1"
from cudaraytracer_amd import capi, scenes
desc = scenes.cat_grid()
for flags in (capi.BUILD_GPU_SAH, capi.BUILD_GPU_SAH, 0):
    t0 = time.perf_counter()
    s = scenes.build(desc, 0, flags)
    print("flags", flags, "total %.1f ms (includes the ctypes add_mesh copies)" % ((time.perf_counter() - t0) * 1e3), "build_ms", s.info().build_ms, flush=True)
