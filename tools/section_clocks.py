"""Section clocks of the instrumented render kernel (CRT_COUNT_WORK + CRT_PRINT_SECTIONS) on a bench workload, for each of the
environment settings given: where a wave's time goes (votes, shade sections, node / leaf steps, ring sums).
usage: python tools/section_clocks.py [cats|cat_grid] ["ENV=1 ENV2=2" ...]"""
import json, os, sys
sys.path.insert(0, '.')
os.environ["CRT_PRINT_SECTIONS"] = "1"
import torch
from cudaraytracer_amd import capi, scenes
name = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].isidentifier() else 'cats'
sets = [a for a in sys.argv[1:] if '=' in a] or [""]
W, H, spp, cam_kw = (3840, 2160, 16, scenes.GRID_CAMERA) if name == 'cat_grid' else (1920, 1080, 64, {})
for env in sets:
    for kv in env.split():
        k, v = kv.split('=')
        os.environ[k] = v
    s = scenes.build(getattr(scenes, name)(), 0)
    _, cam = scenes.make_camera(W, **cam_kw)
    out = torch.empty(W * H * 3, dtype=torch.float32, device='cuda')
    s.render(cam, W, H, spp, out_rgb=out.data_ptr(), sample_chunks=spp, flags=capi.OUT_DEVICE)
    ms = min(s.render(cam, W, H, spp, out_rgb=out.data_ptr(), sample_chunks=spp, flags=capi.OUT_DEVICE)[2].kernel_ms for _ in range(3))
    print("== [%s] %s: %.3f ms plain" % (env, name, ms), flush=True)
    st = s.render(cam, W, H, spp, out_rgb=out.data_ptr(), flags=capi.COUNT_WORK | capi.OUT_DEVICE, sample_chunks=spp)[2]
    sys.stderr.flush()
    print(json.dumps(dict(count_work_ms=round(st.kernel_ms, 2), sched=list(st.sched))), flush=True)
    for kv in env.split():
        os.environ.pop(kv.split('=')[0], None)
