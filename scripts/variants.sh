#!/bin/bash
# usage: scripts/variants.sh SIZE LIB...   -> kbench line per library, two interleaved rounds (same box, same process order)
size=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "product" ]; then python scripts/kbench.py $size 2>/dev/null
    else SIREN_FIT_LIB=$lib python scripts/kbench.py $size 2>/dev/null; fi
  done
done
