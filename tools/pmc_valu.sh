#!/bin/bash
# VALU instruction count of render_kernel<false> + schedule counters, for a list of thresholds
export TMPDIR=/tmp
for T in $1; do
  rm -rf gpurun_out/pv_$T
  CRT_SCHED_THRESHOLD=$T timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/pv_$T -o pv -- python3 tools/sched_stats.py $2 > gpurun_out/pv_$T.log 2>&1 || { echo fail; tail -3 gpurun_out/pv_$T.log; exit 1; }
  python3 - <<PY
import csv, collections, json
rows = list(csv.DictReader(open('gpurun_out/pv_$T/pv_counter_collection.csv')))
agg = collections.defaultdict(list)
for r in rows:
    if '<false' in r['Kernel_Name'] and 'render_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
line = [l for l in open('gpurun_out/pv_$T.log') if l.startswith('{')][-1]
d = json.loads(line)
d.update({k: sum(v)/len(v) for k, v in agg.items()}); d['T'] = $T
print(json.dumps(d))
PY
done
