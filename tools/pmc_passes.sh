#!/bin/bash
# extra PMC passes on tools/sched_stats.py (one launch of the headline frame): stall / issue breakdown
export TMPDIR=/tmp
OUT=gpurun_out/pmcx_${1:-x}; mkdir -p $OUT
pass() { local n=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python3 tools/sched_stats.py > $OUT/$n.log 2>&1 || { echo "$n failed"; tail -3 $OUT/$n.log; exit 1; }; }
pass p1 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_IFETCH
pass p3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
python3 - <<PY
import csv, collections
for n in ['p1','p2','p3']:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open('$OUT/%s/%s_counter_collection.csv' % (n, n))):
        if '<false' in r['Kernel_Name'] and 'render_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items(): print('%-28s %.4g' % (k, sum(v)/len(v)))
PY
