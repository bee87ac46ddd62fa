#!/bin/bash
# wide-path timing of several library builds (product and timing-only), 512x8 on 2048^2, format 12: usage scripts/wide_variants.sh LIB...
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "product" ]; then L=""; else L="$lib"; fi
    SIREN_FIT_LIB=$L python - <<'PY' 2>/dev/null
import os, sys, time, torch
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine, _LIB_PATH
from implicit_image.models import Siren
hidden, depth, size = 512, 8, 2048
eng = SirenEngine(size, size, hidden, depth, compute_dtype="f16", scratch_format=12)
torch.manual_seed(0)
init = Siren(depth=depth, hidden_size=hidden, first_omega_0=50.0, hidden_omega_0=30.0)
eng.set_params(torch.cat([q.detach().reshape(-1) for q in init.parameters()]).cuda())
eng.set_coords(torch.linspace(0, 1, size).cuda(), torch.linspace(0, 1, size).cuda()); eng.set_target(torch.rand(size, size, 3, device="cuda"))
eng.step([3e-4]); torch.cuda.synchronize(); eng.profile(True); eng.profile_reset()
t0 = time.perf_counter(); eng.step([3e-4] * 3); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(os.path.basename(_LIB_PATH), f"{dt*1e3:7.2f} ms/step |", " ".join(f"{k[2:]}={v['total_ms']/3:.2f}" for k, v in eng.profile_report().items() if v['launches'] and v['total_ms']/3 > 0.3))
PY
  done
done
