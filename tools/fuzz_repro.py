"""The two frames of the round-3 parity campaign (tools/fuzz_parity.py, seed 2) that differed from the oracle, rendered on the GPU at their
own parameters and with three node-cache settings: prints ray counts, differing ids and differing image values (all 0 since the
sphere box test).  usage (GPU box): python tools/fuzz_repro.py"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from cudaraytracer_amd import capi, scenes
from oracle import binding as oracle
capi.RELOAD_TUNING_EACH_RENDER = True
desc = scenes.random_spheres()
s = scenes.build(desc, 0)
rs = oracle.RefScene(desc)
cases = [dict(w=191, h=39, spp=6, chunks=0, camera={}, node_cache='3', chunk_mode='1', max_depth=4, seed=813596887249, sample_offset=40, ids_sample=44),
         dict(w=179, h=27, spp=8, chunks=2, camera={'defocus_angle': 0.0}, node_cache='4', chunk_mode='1', max_depth=7, seed=775945998277, sample_offset=33, ids_sample=33)]
for c in cases:
    for nc in (c['node_cache'], '0', '1'):
        os.environ["CRT_NODE_CACHE"] = nc; os.environ["CRT_CHUNK_MODE"] = c['chunk_mode']
        cam_kw = dict(scenes.SPHERES_CAMERA); cam_kw.update(c['camera'])
        params, cam = scenes.make_camera(c['w'], **cam_kw)
        ocam = oracle.make_camera(params, c['w'])
        kw = dict(max_depth=c['max_depth'], seed=c['seed'], sample_offset=c['sample_offset'], ids_sample=c['ids_sample'])
        rgb, ids, st = s.render(cam, c['w'], c['h'], c['spp'], want_ids=True, sample_chunks=0, **kw)
        fb, rids, _, rst = rs.render(ocam, c['w'], c['h'], c['spp'], **kw)
        d = np.argwhere(np.asarray(rgb).reshape(fb.shape) != fb)
        print("case w=%d node_cache=%s: rays %d vs %d, ids differ %d, image differs at %d values" % (c['w'], nc, st.rays, rst.rays, int((ids != rids).sum()), len(d)), flush=True)
        if len(d):
            y, x = int(d[0][0]), int(d[0][1])
            print("   first at pixel (%d,%d): gpu %s oracle %s ids gpu %d oracle %d" % (x, y, np.asarray(rgb).reshape(fb.shape)[y, x], fb[y, x], ids[y, x], rids[y, x]))
