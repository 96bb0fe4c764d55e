"""Schedule and traversal studies on the CPU (no GPU): real per-lane walk traces of the headline frame from the host
single-stepper (tests/hostsim), replayed by tools/sched_sim.cpp under the shipped wave schedule and under candidates, and the
per-pixel entry-point study.  Results are quoted in DESIGN.md section 7.
  python tools/sched_study.py [tile_step]      (every tile_step-th 8x8 tile of the 1920x1080x64 spp cats frame; default 37)"""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cudaraytracer_amd import capi, scenes
from tests import helpers

step = int(sys.argv[1]) if len(sys.argv) > 1 else 37
hs = helpers.HostSim()
hs.L.hostsim_trace_file.argtypes = [C.c_void_p, C.POINTER(capi.Camera), C.POINTER(capi.RenderParams), C.c_uint32, C.c_char_p]
hs.L.hostsim_entry_point_study.argtypes = [C.c_void_p, C.POINTER(capi.Camera), C.POINTER(capi.RenderParams), C.c_uint32, C.c_void_p]
s = hs.build(scenes.cats())
W, H, spp = 1920, 1080, 64
_, cam = scenes.make_camera(W)
p = capi.RenderParams(width=W, height=H, spp=spp, sample_offset=0, max_depth=10, tile_rank=0, tile_ranks=1, flags=0, ids_sample=0, sample_chunks=0, seed=1984, stream=None)
trace = "/tmp/cats_trace.bin"
assert hs.L.hostsim_trace_file(s._h, C.byref(cam), C.byref(p), step, trace.encode()) == 0
sim = "/tmp/sched_sim"
subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(REPO, "tools", "sched_sim.cpp"), "-o", sim])
print("== shipped policy, threshold sweep (the kernel runs 44)")
for t in (32, 40, 44, 48, 56):
    print(subprocess.check_output([sim, trace, "0", str(t)]).decode().strip())
print("== two contexts per lane (one walking, one parked), best refill count per threshold")
for t in (40, 44, 48):
    for r in (8, 16):
        print(subprocess.check_output([sim, trace, "1", str(t), str(r)]).decode().strip())
out = (C.c_uint64 * 8)()
assert hs.L.hostsim_entry_point_study(s._h, C.byref(cam), C.byref(p), step, out) == 0
o = list(out)
print("== per-pixel entry point for the primary rays (ray-exact, i.e. the best any conservative beam test could find)")
print("primary rays %d, walking %.1f %%, node steps per walking ray %.2f, mean entry depth %.2f levels, primary node steps saved %.1f %%, steps added %d"
      % (o[0], 100.0 * o[1] / o[0], o[2] / o[1], o[7] / max(1, o[5] - o[6]), 100.0 * o[3] / o[2], o[4]))
