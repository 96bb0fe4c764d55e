#!/bin/bash
# Interleaved A/B of environment settings on the default bench.py run (two frames in flight, 20 steps), ONE gpurun call:
#   tools/abbench.sh "CRT_NODE_CACHE=0|CRT_NODE_CACHE=1" [rounds]
ROUNDS=${2:-2}
IFS='|' read -ra SETS <<< "$1"
for R in $(seq 1 $ROUNDS); do for E in "${SETS[@]}"; do
  env $E timeout -k 10 120 python -u bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $R [$E]', round(d['value']), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step, kernel alone', round(d['roofline']['kernel_ms'],3))"
done; done
