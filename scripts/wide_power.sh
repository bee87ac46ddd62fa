#!/bin/bash
# rocm-smi power / clocks while the wide step (512x8 on 2048^2, format 12) runs: is it at the package power cap too?  -> gpurun_out/wide_power.txt
out=gpurun_out/wide_power.txt; mkdir -p gpurun_out
python - > gpurun_out/wide_power_run.txt 2>/dev/null <<'PY' &
import os, sys, time, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from implicit_image.models import Siren
hidden, depth, size = 512, 8, 2048
torch.manual_seed(0)
init = torch.cat([q.detach().reshape(-1) for q in Siren(depth=depth, hidden_size=hidden, first_omega_0=50.0, hidden_omega_0=30.0).parameters()]).cuda()
eng = SirenEngine(size, size, hidden, depth, compute_dtype="f16", scratch_format=12)
eng.set_params(init); eng.set_coords(torch.linspace(0, 1, size).cuda(), torch.linspace(0, 1, size).cuda()); eng.set_target(torch.rand(size, size, 3, device="cuda"))
eng.step([3e-4]); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.step([3e-4] * 200); torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / 200 * 1e3:.2f} ms/step", flush=True)
PY
bp=$!
sleep 6
for i in $(seq 1 8); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done > $out
wait $bp
cat gpurun_out/wide_power_run.txt >> $out
cat $out
