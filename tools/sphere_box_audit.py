"""Audit of crt_core.h's sphere_hit_implies_box (the shortcut that spares most sphere hits the reference's literal box test) on rays
made to be hard for it: tangent at the points where a sphere touches its own box, leaving / entering the surface at every angle with
short and long direction vectors, from inside, aimed with impact parameters around the radius, for spheres that are huge, tiny, far
from the world origin or centred on it.  Uses the host stepper (tests/hostsim); a violation = the shortcut said "implied" where the
literal test fails.  `python tools/sphere_box_audit.py [millions of rays per sphere, default 4]`; tests/test_host.py runs a short one."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

SPHERES = [((0.0, -1000.0, 0.0), 1000.0), ((0.0, 0.2, 0.0), 0.2), ((-8.518096, 0.2, 2.8623183), 0.2), ((4.0, 1.0, 0.0), 1.0),
           ((1000.0, 500.0, -2000.0), 0.1), ((3.0, 0.001, -2.0), 0.001), ((0.0, 0.0, 0.0), 1e5), ((0.0, 0.0, 0.0), 1.0),
           ((123.456, -0.5, 77.0), 0.5), ((0.25, 0.5, -0.75), -0.5), ((1e4, 1e4, 1e4), 30.0)]


def rays(center, radius, n, rng):
    """n rays (float32 origins, directions) of the hard kinds, mixed."""
    c = np.array(center, np.float64)
    r = max(abs(radius), 1e-6)
    kind = rng.integers(0, 6, n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    scale = 10.0 ** rng.uniform(-3, 1.7, n)                       # |d| from 1e-3 to 50
    o = np.empty((n, 3))
    d = np.empty((n, 3))
    # 0: tangent at an axis-extreme point, offset by up to +-1e-3 r across the face
    ax = rng.integers(0, 3, n)
    sg = rng.choice([-1.0, 1.0], n)
    e = np.repeat(c[None], n, 0)
    e[np.arange(n), ax] += sg * r
    t = u.copy()
    t[np.arange(n), ax] *= 10.0 ** rng.uniform(-6, -1, n)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    o0 = e - (10.0 ** rng.uniform(-2, 2, n))[:, None] * t * max(r, 1.0)
    o0[np.arange(n), ax] += sg * r * rng.normal(size=n) * 10.0 ** rng.uniform(-8, -3, n)
    # 1: origin on the surface (+- up to 1e-3 r), any direction (outward ones have their far root at 0 +- noise)
    p = c + r * u * (1.0 + rng.normal(size=n)[:, None] * 10.0 ** rng.uniform(-8, -3, n)[:, None])
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    # 2: origin on the surface near an axis-extreme point, direction nearly tangent
    p2 = e + r * 10.0 ** rng.uniform(-5, -1, n)[:, None] * rng.normal(size=(n, 3))
    p2 = c + (p2 - c) / np.linalg.norm(p2 - c, axis=1, keepdims=True) * r * (1.0 + rng.normal(size=n)[:, None] * 1e-6)
    # 3: from inside
    p3 = c + r * u * rng.uniform(0, 1, n)[:, None]
    # 4: aimed with an impact parameter around the radius, from up to 100 r (or 50 units) away
    w = np.cross(u, v)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    impact = r * np.where(rng.random(n) < 0.5, 1.0 + rng.normal(size=n) * 10.0 ** rng.uniform(-7, -2, n), rng.uniform(0, 1.01, n))
    p4 = c + w * impact[:, None] - u * (10.0 ** rng.uniform(-2, 2, n))[:, None] * max(r, 0.5)
    # 5: just outside the box at an axis-extreme point, heading away from the sphere (both roots behind the origin; the far one is
    #    0 +- noise and, with a short direction vector, can exceed tmin)
    p5 = e + r * 10.0 ** rng.uniform(-5, -1.5, n)[:, None] * rng.normal(size=(n, 3))
    p5[np.arange(n), ax] = e[np.arange(n), ax] + sg * r * 10.0 ** rng.uniform(-8, -4, n)
    away = v.copy()
    away[np.arange(n), ax] = sg * np.abs(away[np.arange(n), ax]) * 10.0 ** rng.uniform(-2, 1, n)
    for k, (oo, dd) in enumerate(((o0, t), (p, v), (p2, t), (p3, v), (p4, u), (p5, away))):
        m = kind == k
        o[m] = oo[m]
        d[m] = dd[m]
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d * scale[:, None], np.float32)


def audit(lib, center, radius, millions, seed):
    rng = np.random.default_rng(seed)
    tot = np.zeros(4, np.uint64)
    out = (C.c_uint64 * 4)()
    for _ in range(millions):
        o, d = rays(center, radius, 1_000_000, rng)
        lib.hostsim_sphere_box_audit((C.c_float * 3)(*center), C.c_float(radius), o.ctypes.data, d.ctypes.data, C.c_uint64(len(o)), C.c_float(0.001), out)
        tot += np.array(list(out), np.uint64)
    return tot


def load():
    lib = C.CDLL(os.path.join(REPO, "tests", "hostsim", "libhostsim.so"))
    lib.hostsim_sphere_box_audit.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.POINTER(C.c_uint64)]
    lib.hostsim_sphere_box_audit.restype = None
    return lib


if __name__ == "__main__":
    millions = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    lib = load()
    bad = 0
    for k, (c, r) in enumerate(SPHERES):
        hits, implied, literal, violations = (int(x) for x in audit(lib, c, r, millions, 100 + k))
        print("centre %-28s r %-8g hits %9d  shortcut decides %5.1f %%  literal test passes %7.3f %%  rejected by the box %7d  VIOLATIONS %d"
              % (c, r, hits, 100.0 * implied / max(hits, 1), 100.0 * literal / max(hits, 1), hits - literal, violations), flush=True)
        bad += violations
    sys.exit(1 if bad else 0)
