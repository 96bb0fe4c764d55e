#!/bin/bash
# tools/prof_kernels.sh <tag> [bench args...]: per-kernel time of one bench run (rocprofv3 kernel trace), summary to gpurun_out/<tag>_kernels.txt
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-in-flight 1 "$@" > $OUT/bench.log 2>&1 || exit 1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY' | tee $PWD/gpurun_out/${TAG}_kernels.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s  avg %10.1f us  total %10.3f ms  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
