#!/bin/bash
# same-box A/B of the two forward tilings (k_fwd_pipe: 32x32x16, k_fwd_pipe16: 16x16x32): usage scripts/fwd16_ab.sh [SIZE]
size=${1:-4096}
for round in 1 2; do
  echo -n "32x32x16  "; SIREN_FIT_FWD16=0 python scripts/kbench.py $size 2>/dev/null
  echo -n "16x16x32  "; SIREN_FIT_FWD16=1 python scripts/kbench.py $size 2>/dev/null
done
