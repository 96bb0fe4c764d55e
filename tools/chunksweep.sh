#!/bin/bash
# chunk sweep for a list of libraries: tools/chunksweep.sh "<lib> ..." "<chunks> ..."
for L in $1; do for c in $2; do
  CRT_LIB=$PWD/$L timeout -k 10 100 python bench.py --no-cpu-baseline --chunks $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L chunks $c', round(d['value']), round(d['ms_per_step'],2))"
done; done
