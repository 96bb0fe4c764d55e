#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's default workload on the GPU box.  Usage (from the repo root, via gpurun):
#   bash profiles/run_rocprof.sh <tag>      -> gpurun_out/prof_<tag>/...
# Kernel trace/stats and each PMC pass run separately (never combined), each under its own timeout.
set -u
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --frames-in-flight 1 --no-cpu-baseline ${BENCH_ARGS:-}"   # BENCH_ARGS="--workload cat_grid" profiles another workload
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
# the DEFAULT bench mode (two frames in flight on two streams) under the kernel trace: per-dispatch begin / end timestamps,
# summarised by tools/trace_overlap.py (consecutive dispatches overlap; ends are ms_per_step apart)
BENCH2="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-}"
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace2 -o trace2 -- $BENCH2 > $OUT/trace2.log 2>&1 || { echo "trace2 failed"; tail -5 $OUT/trace2.log; exit 1; }
grep '^{"metric"' $OUT/trace2.log > $OUT/bench_two_in_flight.json || true
grep '^{"metric"' $OUT/trace.log > $OUT/bench.json || true
pass() { # name, counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $BENCH > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; exit 1; }
}
pass pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU
pass pmc_sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass pmc_tcc1 FETCH_SIZE
pass pmc_tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
echo "rocprof passes done"
find $OUT -name '*.csv' | head -40
