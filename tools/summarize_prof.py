"""Summarise a gpurun_out/prof_<tag>/ directory (profiles/run_rocprof.sh) into profiles/<name>/: kernel stats csv + a json
of the per-launch PMC averages of render_kernel<false> with the derived figures DESIGN.md quotes."""
import collections
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
out = {}
for f in ["pmc_sq1", "pmc_sq2", "pmc_tcc1", "pmc_tcc2"]:
    rows = list(csv.DictReader(open(os.path.join(src, f, f + "_counter_collection.csv"))))
    agg = collections.defaultdict(list)
    for r in rows:
        if "kernel<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["LDS_Block_Size"], out["Scratch_Size"], out["Grid_Size"] = r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size"]
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))):
    if "kernel<false" in r["Name"]:
        out["kernel_avg_ns"], out["kernel_calls"] = float(r["AverageNs"]), int(r["Calls"])
cycles = out["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
out["derived"] = {
    "shader_clock_GHz": cycles / out["kernel_avg_ns"],
    "valu_lane_utilisation": out["SQ_THREAD_CYCLES_VALU"] / (out["SQ_ACTIVE_INST_VALU"] * 64.0),
    "valu_busy_fraction_of_simd_time": out["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),   # ACTIVE_INST counts quad-cycles
    "wave_slot_occupancy_of_32_per_cu": out["SQ_WAVE_CYCLES"] * 4.0 / (256.0 * 32.0 * cycles),
    "wave_time_waiting_fraction": out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"],
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
