"""Summarise a gpurun_out/prof_<tag>/ directory (profiles/run_rocprof.sh) into profiles/<name>/: kernel stats csv + a json
of the per-FRAME PMC figures of render_kernel<false> with the derived figures DESIGN.md quotes.

A frame that fills the GPU runs as two render kernels at once (the 16 + 4 split): rocprofv3 serialises
dispatches while it collects counters, so in the PMC passes whichever of the two starts first does the whole frame and the other
finds the work queue empty.  Counters are therefore summed over both kernels of a frame, and the figures quoted are the mean of
the frames that the LARGE kernel rendered (it has 4/5 of the waves in a normal run; the frames that the small kernel rendered --
more node steps from device memory -- are kept beside them as frames_rendered_by_the_small_kernel); the frame's duration comes from the kernel trace, where the two run together
(first start to last end), and the shader clock from the PMC pass's own GRBM_GUI_ACTIVE and timestamps."""
import collections
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
import re
RENDER = re.compile(r"render_kernel\w*<false")                                   # the timed (not instrumented) render kernels
LARGE = re.compile(r"render_kernel\w*<false, \d+, (true|false), 16,")          # ... with 16 waves per workgroup
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
out = {}
for f in ["pmc_sq1", "pmc_sq2", "pmc_tcc1", "pmc_tcc2"]:
    rows = list(csv.DictReader(open(os.path.join(src, f, f + "_counter_collection.csv"))))
    # dispatches of the render kernels in start order; with two kernels per frame, consecutive pairs are frames
    disp_rows = collections.OrderedDict()
    for r in rows:
        if RENDER.search(r["Kernel_Name"]):
            d = disp_rows.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"]), "c": {}})
            d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            out.setdefault("kernels", {})[r["Kernel_Name"].split("(")[0]] = {"LDS_Block_Size": r["LDS_Block_Size"], "Scratch_Size": r["Scratch_Size"],
                                                                          "Grid_Size": r["Grid_Size"], "VGPR_Count": r["VGPR_Count"]}
    ds = sorted(disp_rows.values(), key=lambda d: d["start"])
    per_frame = 2 if len(set(d["name"] for d in ds)) == 2 else 1
    frames_p = [ds[k:k + per_frame] for k in range(0, len(ds), per_frame)]
    # which kernel rendered the frame (the PMC passes serialise the two): the one that ran longer
    kinds = collections.defaultdict(list)
    for fr in frames_p:
        worker = max(fr, key=lambda d: d["end"] - d["start"])
        kinds["large_kernel_first" if LARGE.search(worker["name"]) or per_frame == 1 else "small_kernel_first"].append(fr)
    main = kinds["large_kernel_first"] or kinds["small_kernel_first"]
    def mean(frs, key):
        return sum(sum(d["c"].get(key, 0.0) for d in fr) for fr in frs) / len(frs)
    names = sorted(set(k for d in ds for k in d["c"]))
    for k in names:
        out[k] = mean(main, k)
    if kinds["small_kernel_first"] and kinds["large_kernel_first"]:
        o2 = out.setdefault("frames_rendered_by_the_small_kernel", {})
        for k in names:
            o2[k] = mean(kinds["small_kernel_first"], k)
    out.setdefault("pmc_frames", {})[f] = {k: len(v) for k, v in kinds.items()}
    if "GRBM_GUI_ACTIVE" in names:
        out["pmc_pass_ns_per_frame"] = sum(sum(d["end"] - d["start"] for d in fr) for fr in main) / len(main)
# the frame's duration where its kernels run together: the kernel trace, first start to last end of each frame
disp = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in
               csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))) if RENDER.search(r["Kernel_Name"])))
frames_t = []
for s0, e0, name in disp:
    if frames_t and s0 < frames_t[-1][1] and name not in frames_t[-1][2]:
        frames_t[-1][1] = max(frames_t[-1][1], e0); frames_t[-1][2].add(name)
    else:
        frames_t.append([s0, e0, {name}])
out["kernel_avg_ns"], out["kernel_calls"] = sum(e - s for s, e, _ in frames_t) / len(frames_t), len(frames_t)
out["kernels_per_frame"] = len(frames_t[0][2])
clock_ghz = out["GRBM_GUI_ACTIVE"] / 8.0 / out["pmc_pass_ns_per_frame"]  # summed over the 8 XCDs
cycles = clock_ghz * out["kernel_avg_ns"]
out["derived"] = {
    "shader_clock_GHz": clock_ghz,
    "valu_lane_utilisation": out["SQ_THREAD_CYCLES_VALU"] / (out["SQ_ACTIVE_INST_VALU"] * 64.0),
    "valu_busy_fraction_of_simd_time": out["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),   # ACTIVE_INST counts quad-cycles
    "wave_time_waiting_fraction_serialised": out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"],
    "hbm_bytes_per_launch": out["FETCH_SIZE"] * 1024.0 * 2.0 + out["WRITE_SIZE"] * 1024.0,          # FETCH_SIZE x2: gfx950 calibration
    "l2_hit_rate": out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]),
    "valu_instructions_per_launch": out["SQ_INSTS_VALU"],
}
for name in ("bench.json", "bench_two_in_flight.json"):
    if os.path.isfile(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
trace2 = os.path.join(src, "trace2", "trace2_kernel_trace.csv")
if os.path.isfile(trace2) and os.path.isfile(os.path.join(src, "bench_two_in_flight.json")):
    import subprocess
    with open(os.path.join(dst, "two_in_flight_trace.json"), "w") as f:
        subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "trace_overlap.py"), trace2,
                               os.path.join(src, "bench_two_in_flight.json")], stdout=f)
json.dump(out, open(os.path.join(dst, "pmc_render_kernel.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
