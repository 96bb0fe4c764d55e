"""Reads a rocprofv3 --kernel-trace CSV of `bench.py` (default mode: two frames in flight on two streams) and reports, for
the render kernel's dispatches of the timed region, how consecutive dispatches overlap and how far apart they END --
the per-step kernel figure that must agree with the bench line's ms_per_step.
  python tools/trace_overlap.py <kernel_trace.csv> [bench.json] > profiles/<dir>/two_in_flight_trace.json"""
import csv
import re
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
bench = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None


def col(r, *names):
    for n in names:
        if n in r:
            return r[n]
    raise KeyError(names)


disp = []
for r in rows:
    name = col(r, "Kernel_Name", "Name")
    if re.search(r"render_kernel\w*<false", name):
        disp.append((int(col(r, "Start_Timestamp", "BeginNs")), int(col(r, "End_Timestamp", "EndNs")), col(r, "Queue_Id") if "Queue_Id" in r else "?"))
disp.sort()
# A frame that fills the GPU runs as two kernels at once (the 16 + 4 split: a 1024-thread and a 256-thread variant): the k-th
# dispatch of one kind and the k-th of the other are one frame, from the first start to the last end.
kinds = {}
for r in rows:
    name = col(r, "Kernel_Name", "Name")
    if re.search(r"render_kernel\w*<false", name):
        kinds.setdefault(name, []).append((int(col(r, "Start_Timestamp", "BeginNs")), int(col(r, "End_Timestamp", "EndNs")), col(r, "Queue_Id") if "Queue_Id" in r else "?"))
kernels_per_frame = 1
if len(kinds) == 2 and len(set(len(v) for v in kinds.values())) == 1:
    a, b = (sorted(v) for v in kinds.values())
    disp = sorted((min(x[0], y[0]), max(x[1], y[1]), x[2] + "+" + y[2]) for x, y in zip(a, b))
    kernels_per_frame = 2
steps = bench["steps"] if bench else None
warm = bench["warmup"] if bench else 0
# the timed region: `steps` dispatches after the warm-up ones; the three stand-alone launches and the instrumented one follow
timed = disp[warm:warm + steps] if steps else disp
dur = [(e - s) / 1e6 for s, e, _ in timed]
end_gap = [(timed[k + 1][1] - timed[k][1]) / 1e6 for k in range(len(timed) - 1)]
start_gap = [(timed[k + 1][0] - timed[k][0]) / 1e6 for k in range(len(timed) - 1)]
overlap = [max(0.0, (timed[k][1] - timed[k + 1][0]) / 1e6) for k in range(len(timed) - 1)]
alone = disp[warm + steps:warm + steps + 3] if steps else []
out = {
    "kernels_per_frame": kernels_per_frame, "render_frames_total": len(disp), "timed_dispatches": len(timed),
    "queues_of_timed_frames": sorted(set(q for _, _, q in timed)),
    "frame_duration_ms_mean": sum(dur) / len(dur), "frame_duration_ms_min": min(dur), "frame_duration_ms_max": max(dur),
    "end_to_end_ms_mean": sum(end_gap) / len(end_gap), "start_to_start_ms_mean": sum(start_gap) / len(start_gap),
    "overlap_with_next_ms_mean": sum(overlap) / len(overlap), "frames_overlapping_the_next": sum(1 for o in overlap if o > 0),
    "standalone_frame_ms": [(e - s) / 1e6 for s, e, _ in alone],
}
if bench:
    out["bench_ms_per_step"] = bench["ms_per_step"]
    out["bench_per_step_kernel_end_to_end_ms"] = bench["roofline"].get("per_step", {}).get("kernel_end_to_end_ms")
    out["bench_kernel_ms_standalone"] = bench["roofline"]["kernel_ms"]
    out["agrees"] = abs(out["end_to_end_ms_mean"] - bench["ms_per_step"]) / bench["ms_per_step"] < 0.05
print(json.dumps(out, indent=1))
