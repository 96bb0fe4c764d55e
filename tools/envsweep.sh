#!/bin/bash
# tools/envsweep.sh VAR "v1 v2 ..."  : bench with an environment knob
for V in $2; do
  env $1=$V timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1=$V', round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms nodes/ray %.2f tris/ray %.2f'%(r['nodes_per_ray'], r['tri_tests_per_ray']))"
done
