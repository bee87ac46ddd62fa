#!/usr/bin/env python3
"""Launch-bound regime: steps/s of small fits (BASELINE config 1: SIREN 64x4 on 256x256) through sf_step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "implicit-image-compression_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from implicit_image._engine import SirenEngine  # noqa: E402
from implicit_image.models import Siren  # noqa: E402

for hidden, depth, size, n in ((64, 4, 256, 2000), (128, 8, 256, 2000), (256, 8, 512, 1000)):
    eng = SirenEngine(size, size, hidden, depth, compute_dtype="f16")
    torch.manual_seed(0)
    init = Siren(depth=depth, hidden_size=hidden, first_omega_0=50.0, hidden_omega_0=30.0)
    dev = torch.device("cuda")
    eng.set_params(torch.cat([q.detach().reshape(-1) for q in init.parameters()]).to(dev))
    eng.set_coords(torch.linspace(0, 1, size).to(dev), torch.linspace(0, 1, size).to(dev))
    eng.set_target(torch.rand(size, size, 3, device=dev))
    eng.step([3e-4] * 50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step([3e-4] * n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{hidden}x{depth} @ {size}^2: {dt * 1e6:.1f} us/step, {size * size / dt / 1e6:.1f} Mpix-iters/s", flush=True)
    eng.close()
