#!/bin/bash
# ring-mode parameter sweep on the headline frame (CRT_CHUNK_MODE=2 forces the rings)
S=""
for K in 4 8 16; do for B in 128 256; do for N in 1 2 4 8; do
  if [ $N -le $((K/2)) ]; then S="$S|CRT_CHUNK_MODE=2 CRT_RING_SLOTS=$K CRT_FETCH_BLOCK=$B CRT_RING_BATCH=$N"; fi
done; done; done
S="CRT_CHUNK_MODE=1${S}|CRT_CHUNK_MODE=2 CRT_RING_SLOTS=16 CRT_FETCH_BLOCK=512 CRT_RING_BATCH=4|CRT_CHUNK_MODE=2 CRT_RING_SLOTS=16 CRT_FETCH_BLOCK=512 CRT_RING_BATCH=8"
STEPS=10 bash tools/abenv.sh "$S" 1
