#!/usr/bin/env python3
"""Per-kernel times of one small fit step (config 1 shape) from the engine's HIP-event profiler."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "implicit-image-compression_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from implicit_image._engine import SirenEngine  # noqa: E402

for hidden, depth, size in ((64, 4, 256), (256, 8, 512)):
    eng = SirenEngine(size, size, hidden, depth, compute_dtype="f16")
    dev = torch.device("cuda")
    eng.set_params(torch.randn(eng.num_params, device=dev) * 0.01)
    eng.set_coords(torch.linspace(0, 1, size).to(dev), torch.linspace(0, 1, size).to(dev))
    eng.set_target(torch.rand(size, size, 3, device=dev))
    eng.step([3e-4] * 3)
    eng.profile(True)
    eng.profile_reset()
    n = 20
    eng.step([3e-4] * n)
    rep = eng.profile_report()
    print(hidden, depth, size, {k: (round(v["total_ms"] / n * 1e3, 1), v["launches"] // n) for k, v in rep.items() if v["launches"]},
          "sum_us", round(sum(v["total_ms"] for v in rep.values()) / n * 1e3, 1), flush=True)
    eng.close()
