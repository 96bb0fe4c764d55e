"""bench.py -- headline benchmark: Mrays/s (all bounces) + ms/frame, cats.obj 1920x1080 64 spp (BASELINE.json).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode tile|sample] [--workload cats|cat_checker|cat_grid]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full frame of the workload.  N ranks split the SAME frame (strong scaling): tile mode deals 8x8 tiles
round-robin and all-gathers them over RCCL; sample mode splits the samples and reduces float sums.  The scene, camera
and output buffers are resident in HBM before the timed region; `value` = path segments traced by all ranks / wall time.
Rank 0 prints ONE JSON line.  `roofline` is for the render kernel (VALU-issue bound: no dense contraction, scene
L2-resident; HBM figures given beside it); `cpu_baseline` times the oracle (CPU restatement of the reference) on the
host cores on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import re
import sys
import time

# Two render streams, an epilogue stream, the collective's stream and the default stream: more than HIP's default of 4
# hardware queues per process, and streams that share a queue run one after the other (measured on 1/8 of the frame with a
# 1-rank RCCL group, tools/pipeline_check.py: 1.80 ms per step with 4 queues, 1.54 ms with 8).  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # before the HIP library: one HIP runtime per process (see tests/conftest.py)
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from cudaraytracer_amd import capi, scenes  # noqa: E402
from cudaraytracer_amd.dist import SampleFrame, TileFrame  # noqa: E402

WORKLOADS = {
    # name: (scene builder, width, height, spp, max_depth, camera overrides)
    "cats": ("cats", 1920, 1080, 64, 10, {}),            # BASELINE configs[2] = the config the metric is quoted on
    "cat_checker": ("cat_checker", 1920, 1080, 64, 8, {}),  # configs[1]
    "cat_grid": ("cat_grid", 3840, 2160, 16, 10, scenes.GRID_CAMERA),  # configs[3]
    "cats_4k": ("cats", 3840, 2160, 1024, 10, {}),       # configs[4] (use --mode sample)
    "spheres": ("random_spheres", 1920, 1080, 64, 10, scenes.SPHERES_CAMERA),  # row N3: the reference's sphere scene (scene.h:53-90)
}

# /opt/skills/guides/MI355X_MICROARCH.md: 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz (= the 157.3 TFLOP/s fp32 spec / 2)
VALU_PEAK_TLANEOPS_256CU = 256 * 4 * 32 * 2.4e9 / 1e12
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0          # aggregate L2, same guide
INFINITY_CACHE_GATHER_GBS = 8600.0  # random 1 KB-row gathers from a 38 MB table (Infinity Cache resident), same guide
# Default: one sample per work unit, whatever the workload's sample count: the samples of a pixel are then added singly in sample
# order, which is the reference's own loop (renderer.h:26-32), so the timed frame is bit-identical to the whole-pixel one
# (tests/test_gpu.py::test_one_sample_per_chunk_is_the_sequential_sum, also at the full 1080p x 64 spp).  Same for every N.
# (The library decides per frame whether the units' results go through slabs or the waves' rings: crt_api.cpp, plan_chunks.)


def lane_ops(st, node_width=2):
    # 30 per box test (2 per node; 4 with CRT_BVH_WIDE=1), 60 per triangle test, 25 per sphere test, 120 per shaded bounce
    return 30.0 * node_width * st.nodes + 60.0 * st.tri_tests + 25.0 * st.sphere_tests + 120.0 * st.shaded_hits


def algorithmic_bytes(st, width, height, node_width=2):
    return 32.0 * node_width * st.nodes + 48.0 * st.tri_tests + 48.0 * st.sphere_tests + 12.0 * width * height


def pmc_traffic(workload, world):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/run_rocprof.sh: FETCH_SIZE and WRITE_SIZE in their own runs; tools/summarize_prof.py applies the guide's
    unit and gfx950 corrections): the newest profiles/<dir>/pmc_render_kernel.json whose bench line was this workload on
    one GPU, and where it came from.  (None, None) when there is none."""
    if world != 1:
        return None, None
    root = os.path.join(REPO, "profiles")
    best = None
    def natural(name):  # r2_v10_... after r2_v9_...
        return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]
    for name in sorted(os.listdir(root), key=natural) if os.path.isdir(root) else []:
        f = os.path.join(root, name, "pmc_render_kernel.json")
        b = os.path.join(root, name, "bench.json")
        if not (os.path.isfile(f) and os.path.isfile(b)):
            continue
        try:
            cfg = json.load(open(b))
            if cfg.get("n_gpus") == 1 and cfg["config"]["workload"].split()[0] == workload:
                best = f
        except (KeyError, ValueError, OSError, IndexError):
            continue
    if not best:
        return None, None
    try:
        return float(json.load(open(best))["derived"]["hbm_bytes_per_launch"]), "profiles/%s (committed PMC pass)" % os.path.basename(os.path.dirname(best))
    except (KeyError, ValueError, OSError):
        return None, None


def cpu_baseline(desc, cam_params, width, height, max_depth, seed, target_seconds=4.0, reps=3):
    """Oracle (oracle/crt_ref.c, reference traversal semantics) on the host cores: whole frame, a few samples, median of
    `reps` repetitions; also the reference traversal's own work per ray (48-byte nodes, one box per node, no culling)."""
    from oracle import binding as ob
    rs = ob.RefScene(desc)
    cam = ob.make_camera(cam_params, width)
    cores = min(os.cpu_count() or 1, 16)
    t0 = time.perf_counter()
    _, _, _, st = rs.render(cam, width, height, 1, max_depth=max_depth, seed=seed, nthreads=cores, want_ids=False)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(64, target_seconds / max(t1, 1e-3))))
    rates, secs = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        _, _, _, st = rs.render(cam, width, height, spp, max_depth=max_depth, seed=seed, nthreads=cores, want_ids=False)
        dt = time.perf_counter() - t0
        rates.append(st.rays / dt / 1e6)
        secs.append(dt)
    rates.sort()
    return dict(value=rates[len(rates) // 2], unit="Mrays/s", cores=cores, kind="port",
                sample="whole %dx%d frame, samples 0..%d of the workload (%d rays per repetition, median of %d repetitions of %.1f s, %d threads over rows)"
                       % (width, height, spp - 1, st.rays, reps, sorted(secs)[len(secs) // 2], cores),
                reference_traversal={"box_tests_per_ray": st.box_tests / max(st.rays, 1), "triangle_tests_per_ray": st.tri_tests / max(st.rays, 1),
                                     "sphere_tests_per_ray": st.sphere_tests / max(st.rays, 1)})


def image_note(chunks, unit_spp, sample_split=False):
    """What the timed frame is, against the reference's sequential sample-order sum (renderer.h:26-32)."""
    if sample_split:
        return ("sample split: every rank sums its own samples, the ranks' sums are added in rank order; same samples as the reference's "
                "sequential sum, differs from it by float association only (rtol 3e-7 on the gamma-encoded value, tests/test_dist.py)")
    if chunks <= 1:
        return "bit-exact: every pixel's samples summed in sample order (renderer.h:28-32)"
    if chunks == unit_spp:
        return ("bit-exact: one sample per work unit, the %d single-sample slabs of a pixel added in sample order starting from 0 -- the float "
                "operations of the reference's loop (renderer.h:26-32); equal to the whole-pixel frame bit for bit, tested at this size" % chunks)
    return ("chunked-sum variant: same samples, each pixel's %d chunk sums added in chunk order; differs from the reference's sequential "
            "sample-order sum by float association only (<= 3e-7 relative, tests/test_gpu.py); --chunks 1 and the default (one sample per "
            "unit) are bit-exact" % chunks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-in-flight", type=int, choices=[0, 1, 2, 3], default=0,
                    help="n > 1: consecutive frames rotate over n HIP streams (dist.py), so one frame's last waves drain while the next "
                         "frame's first waves start; 1: one frame at a time; 0 (default): 2.  (A third frame in flight helped short "
                         "rank-frames by 2-3 %% while a CU held five 4-wave workgroups; with one 16-wave and one 4-wave workgroup per CU "
                         "it costs 5 %%: tools/rank_share.py, DESIGN.md section 5)")
    ap.add_argument("--mode", choices=["tile", "sample"], default="tile")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cats")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debug; invalidates the headline number)")
    ap.add_argument("--chunks", type=int, default=-1,
                    help="sample chunks per pixel (crt_render_params.sample_chunks): work unit = (pixel, spp/chunks samples); "
                         "1 = whole pixel per lane; default: one sample per unit (= spp); both are the bit-exact sample-order sum; "
                         "any other value is lowered to a divisor of spp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if os.environ.get("CRT_BENCH_WATCHDOG"):  # where every thread stands after so many seconds (rehearsals of the N-rank path)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CRT_BENCH_WATCHDOG"]), repeat=False, file=sys.stderr)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    in_flight = args.frames_in_flight or 2
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no HIP device visible; the hot path has no CPU fallback")
    # Rehearsal on a one-GPU box (not a measurement): CRT_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and
    # CRT_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); the N-rank code path -- launch, tile deal,
    # pipelined gather / reduce on the side streams, barrier, max over ranks, the JSON line -- then runs as on an 8-GPU node.
    share = os.environ.get("CRT_BENCH_SHARE_DEVICE") == "1"
    backend = os.environ.get("CRT_BENCH_BACKEND", "nccl")
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    builder, W, H, spp, depth, cam_over = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    seed = scenes.REF_SEED
    desc = getattr(scenes, builder)()
    scene = scenes.build(desc, local_rank)
    cam_params, cam = scenes.make_camera(W, **cam_over)
    info = scene.info()

    chunks = spp if args.chunks < 0 else min(max(args.chunks, 1), spp)
    while spp % chunks:
        chunks -= 1
    if args.mode == "tile":
        frame = TileFrame(W, H, world, rank, device, sample_chunks=chunks, pipelined=in_flight if in_flight > 1 else False)

        def launch(want_stats=False, flags=0):
            return frame.render_local(scene, cam, spp, depth, seed, want_stats=want_stats, flags=flags)
        finish = frame.gather_async  # two frames in flight on two streams (dist.py): drain() completes the last ones
    else:
        frame = SampleFrame(W, H, spp, world, rank, device, sample_chunks=chunks, pipelined=in_flight if in_flight > 1 else False)

        def launch(want_stats=False, flags=0):
            return frame.render_local(scene, cam, depth, seed, want_stats=want_stats, flags=flags)
        finish = frame.reduce_async

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        launch()
        finish()
    frame.drain()
    barrier()
    # Nothing inside the timed region waits on the host.  An event behind each frame's render (on its render stream) gives
    # the per-step kernel figure: with two frames in flight a dispatch's own begin-to-end time includes its wait for the
    # previous frame's wave slots, what counts is how far apart consecutive frames END.
    ends = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        launch()
        ev = torch.cuda.Event(enable_timing=True)
        cur = frame._current[1] if frame._current is not None else None
        ev.record(cur if cur is not None else torch.cuda.current_stream(device))
        ends.append(ev)
        finish()
    frame.drain()  # every one of the K frames is assembled (tile: on every rank; sample: on rank 0) inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    # The kernel's own launch duration for the roofline: HIP events on the launch stream around the kernel (recorded by
    # the library, crt_stats.kernel_ms), over launches that run alone -- with two frames in flight a kernel's start-to-end
    # time includes the wait for the wave slots of the frame before it, which says nothing about the kernel.
    kernel_ms, rays_local = [], 0
    for _ in range(3):
        st = launch(want_stats=True)
        finish()
        frame.drain()
        kernel_ms.append(st.kernel_ms)
        rays_local = st.rays  # deterministic: the same frame again

    # mean distance between the ends of consecutive frames' render work (kernel + chunk-sum pass) inside the timed region
    end_to_end_ms = (ends[0].elapsed_time(ends[-1]) / (len(ends) - 1)) if len(ends) > 1 else float("nan")
    t = torch.tensor([elapsed, float(rays_local), sum(kernel_ms) / max(len(kernel_ms), 1), end_to_end_ms], dtype=torch.float64, device=device)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, rays_frame, kernel_avg_ms, end_to_end_ms = float(tmax[0]), float(tsum[1]), float(tmax[2]), float(tmax[3])
    else:
        rays_frame, kernel_avg_ms = float(rays_local), float(t[2])

    # work counts of this rank's launch (instrumented kernel, outside the timed region; counts are deterministic)
    stw = launch(want_stats=True, flags=capi.COUNT_WORK)
    finish()
    frame.drain()
    torch.cuda.synchronize()

    if rank == 0:
        props = torch.cuda.get_device_properties(device)
        cus = props.multi_processor_count
        valu_peak = VALU_PEAK_TLANEOPS_256CU * cus / 256.0
        ksec = kernel_avg_ms / 1e3
        ops = lane_ops(stw, info.node_width)
        traffic, traffic_source = pmc_traffic(args.workload, world)
        abytes = algorithmic_bytes(stw, W, H, info.node_width)
        used_chunks = chunks if args.mode == "tile" else frame.sample_chunks
        out = {
            "metric": "Mrays/s (all bounces)", "value": rays_frame * args.steps / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d %dspp depth %d (models/%s + ground + emissive sphere, seed %d)" %
                       (args.workload, W, H, spp, depth, {"cats": "cats.obj", "random_spheres": "no mesh"}.get(builder, "cat.obj"), seed),
                       "partition": "%s x%d%s" % (args.mode, world, " (REHEARSAL: ranks share device 0 over %s, not a measurement)" % backend if share else ""), "frames_in_flight": frame.slots.in_flight, "sample_chunks": used_chunks,
                       "image": image_note(used_chunks, spp if args.mode == "tile" else spp // world, args.mode == "sample" and world > 1),
                       "triangles": info.num_triangles, "bvh_nodes": info.num_nodes, "bvh_depth": info.bvh_depth, "bvh_node_width": info.node_width,
                       "rays_per_frame": int(rays_frame), "device_cus": cus},
            "roofline": {"bound": "valu", "achieved": ops / ksec / 1e12, "peak": valu_peak, "unit": "Tlane-op/s",
                         "frac": ops / ksec / 1e12 / valu_peak, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "render_kernel", "kernel_ms": kernel_avg_ms, "kernel_ms_note": "mean of 3 launches that run alone, right after the timed region",
                         "per_step": {"kernel_end_to_end_ms": end_to_end_ms, "frac": ops / (end_to_end_ms / 1e3) / 1e12 / valu_peak,
                                      "note": "mean distance between the ends of consecutive frames' render work in the timed region (HIP events on the "
                                              "render streams): what a frame costs with %d frames in flight; <= ms_per_step" % frame.slots.in_flight},
                         "rays_per_launch": int(stw.rays), "lane_ops_per_launch": ops,
                         "nodes_per_ray": stw.nodes / max(stw.rays, 1), "tri_tests_per_ray": stw.tri_tests / max(stw.rays, 1),
                         "sphere_tests_per_ray": stw.sphere_tests / max(stw.rays, 1), "shaded_hits_per_ray": stw.shaded_hits / max(stw.rays, 1),
                         "hbm": {"bound": "hbm", "achieved": (traffic / ksec / 1e9) if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (traffic / ksec / 1e9 / HBM_PEAK_GBS) if traffic else None,
                                 "algorithmic_bytes_per_launch": abytes,
                                 "note": "achieved = PMC HBM bytes per launch (traffic) / kernel time; the algorithmic bytes are priced in l2_lds"},
                         "l2_lds": {"bound": "l2", "achieved": abytes / ksec / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": abytes / ksec / 1e9 / L2_PEAK_GBS,
                                    "infinity_cache_gather_peak": INFINITY_CACHE_GATHER_GBS, "frac_of_infinity_cache_gather": abytes / ksec / 1e9 / INFINITY_CACHE_GATHER_GBS,
                                    "note": "algorithmic node (64 B; 128 B with four children) + triangle (48 B) + sphere (48 B) + framebuffer bytes / kernel time: a request rate served by the "
                                            "LDS node copy, L1 and L2 when the scene (%d KiB) is cache resident; for scenes beyond L2 (the 1 M-triangle grid) the "
                                            "Infinity-Cache gather rate is the memory-side bound"
                                            % ((info.node_bytes + info.triangle_bytes + info.sphere_bytes) // 1024)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, cam_params, W, H, depth, seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
