#!/bin/bash
# k_fwd time per step (4 launches of 4 Mi pixels) for the product library and for timing-only builds in build/x_*.so
# (results of those are wrong by construction; only the clock is read).  usage: scripts/fwd_variants.sh [libs...]
for lib in "" "$@"; do
  SIREN_FIT_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${lib:-product}', 'step %.2f ms' % d['ms_per_step'], {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items() if k in ('k_fwd','k_bwd_hidden')})"
done
