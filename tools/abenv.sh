#!/bin/bash
# Interleaved A/B of environment settings (same library) in ONE gpurun call:
#   tools/abenv.sh "CRT_RASTER_ORDER=1|CRT_RASTER_ORDER=0|CRT_SCHED_THRESHOLD=32 CRT_WAVES_PER_CU=20" [rounds]
ROUNDS=${2:-2}
IFS='|' read -ra SETS <<< "$1"
for R in $(seq 1 $ROUNDS); do for E in "${SETS[@]}"; do
  env $E timeout -k 10 120 python -u bench.py --steps ${STEPS:-3} --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $R [$E]', round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms')"
done; done
