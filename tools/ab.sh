#!/bin/bash
# Interleaved A/B of library builds in ONE process group on ONE box (devices differ by ~10 %: never compare across gpurun calls)
#   tools/ab.sh "<lib1> <lib2> ..." [rounds]
ROUNDS=${2:-3}
for R in $(seq 1 $ROUNDS); do for L in $1; do
  CRT_LIB=$PWD/$L timeout -k 10 120 python -u bench.py --steps ${STEPS:-20} --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $R $L', round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms')"
done; done
