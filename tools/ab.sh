#!/bin/bash
# A/B bench of library variants:  tools/ab.sh "<lib1> <lib2> ..." "<threshold list>"
for L in $1; do for T in $2; do
  CRT_LIB=$PWD/$L CRT_SCHED_THRESHOLD=$T timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L T=$T', round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms')"
done; done
