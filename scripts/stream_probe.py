#!/usr/bin/env python3
"""Probe: step rate of the same fit on the default (null) stream vs an explicitly created stream, and with
SIREN-initialised vs small random weights (clock / data effects)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "implicit-image-compression_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from implicit_image._engine import SirenEngine  # noqa: E402
from implicit_image.models import Siren  # noqa: E402

size, steps = 4096, 5
dev = torch.device("cuda")
torch.manual_seed(0)
init = torch.cat([q.detach().reshape(-1) for q in Siren(depth=8, hidden_size=256, first_omega_0=50., hidden_omega_0=30.).parameters()]).to(dev)
img = torch.rand(size, size, 3, device=dev)
for name, stream in (("null stream", None), ("created stream", torch.cuda.Stream())):
    for wname, w in (("siren init", init), ("uniform +-0.03", (torch.rand_like(init) * 2 - 1) * 0.03)):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            e = SirenEngine(size, size, 256, 8, compute_dtype="f16")
            e.set_params(w.clone()); e.set_coords(torch.linspace(0, 1, size).to(dev), torch.linspace(0, 1, size).to(dev))
            e.set_target(img)
            e.step([3e-4] * 2); torch.cuda.synchronize()
            t0 = time.perf_counter(); e.step([3e-4] * steps); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
            print(f"{name:15s} {wname:15s}: {dt * 1e3:.2f} ms/step, {size * size / dt / 1e6:.1f} Mpix-it/s", flush=True)
            e.close()
