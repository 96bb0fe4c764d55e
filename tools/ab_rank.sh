#!/bin/bash
# Interleaved A/B of library builds on the N-way partition cost (tools/rank_share.py): tools/ab_rank.sh "<lib1> <lib2>" "<N> <chunks>" [rounds]
ROUNDS=${3:-2}
for R in $(seq 1 $ROUNDS); do for L in $1; do
  echo "round $R $L: $(CRT_LIB=$PWD/$L timeout -k 10 200 python tools/rank_share.py $2 2>/dev/null | tr '\n' ' ')"
done; done
