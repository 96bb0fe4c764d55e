"""Probe of a parity case DESIGN.md section 2 describes for spheres IN THE TREE (scenes with more than 8 spheres): 15 spheres of
radius 0.2 hovering HOVER (default 2e-4) above a two-triangle floor, rays from BELOW the floor aimed at the points under the
spheres.  The reference does not cull by distance, so it takes the sphere whenever the sphere's computed root -- which carries the
discriminant's cancellation, ~1e-4..1e-3 of t from 30 units away -- comes out smaller than the floor's t.  With the triangle walk's
pruning limit (best + 1e-6) the product had the floor first, pruned the sphere's node and answered the floor for ~10 % of these rays;
with crt_traverse.h's prune_limit for sphere-slot scenes: none.  Host stepper against the oracle; prints the number of rays that differ.
usage: [HOVER=2e-4] python tools/sphere_prune_probe.py [rays]"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cudaraytracer_amd import capi, scenes
from oracle import binding as oracle
import helpers
hs = helpers.HostSim()
desc = scenes._base()
desc["materials"].append(dict(type=capi.MAT_LAMBERTIAN, rgb=(0.5, 0.5, 0.5), param=0.0, texture=None))
rng = np.random.default_rng(1)
cs = [(float(x), 0.2 + float(os.environ.get("HOVER", "2e-5")), float(z)) for x in range(-2, 3) for z in range(-1, 2)]
desc["spheres"] = [dict(center=c, radius=0.2, material=0) for c in cs]
v = np.array([[-50, 0, -50], [50, 0, -50], [50, 0, 50], [-50, 0, 50]], np.float32)
desc["meshes"].append(dict(vertices=v, indices=np.array([[0, 1, 2], [0, 2, 3]], np.uint32), material=0))
s = hs.build(desc); rs = oracle.RefScene(desc)
print("spheres in tree:", s.info().num_spheres, "sphere_bytes", s.info().sphere_bytes)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
mism = 0
for k in range(n):
    c = np.array(cs[rng.integers(len(cs))]); 
    target = np.array([c[0], 0.0, c[2]]) + np.array([rng.normal(), 0, rng.normal()]) * 0.2 * 10.0 ** rng.uniform(-4, -1)
    az = rng.uniform(0, 2 * np.pi); el = 10.0 ** rng.uniform(-2, 0.19)
    u = np.array([np.cos(az) * np.cos(el), -np.sin(el), np.sin(az) * np.cos(el)])   # origin below the floor
    L = 10.0 ** rng.uniform(0, 2)
    o = (target + L * u).astype(np.float32)
    d = (-u * 10.0 ** rng.uniform(-1, 1.3)).astype(np.float32)
    ref = rs.trace(o, d)
    p, t, took = hs.trace(s, o, d)
    if p != ref[0] or (p >= 0 and t != ref[1]):
        mism += 1
        if mism <= 3: print("MISMATCH", o, d, (p, t), ref)
print("rays from below the floor at the contact points:", n, "mismatches", mism)
