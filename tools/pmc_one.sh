#!/bin/bash
# one PMC pass over tools/sched_stats.py (one headline frame per launch): tools/pmc_one.sh <tag> <counter> [<counter> ...]
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc1_$TAG; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/p -o p -- python3 tools/sched_stats.py > $OUT/p.log 2>&1 || { echo "pass failed"; tail -3 $OUT/p.log; exit 1; }
python3 - <<PY
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open('$OUT/p/p_counter_collection.csv')):
    if '<false' in r['Kernel_Name'] and 'render_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items(): print('$TAG %-32s %.4g' % (k, sum(v)/len(v)))
PY
