#!/usr/bin/env python3
"""Wide-path timing: per-kernel table of one fit step at hidden 512 / 1024 (BASELINE configs 3 and 5 shapes)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "implicit-image-compression_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from implicit_image._engine import SirenEngine  # noqa: E402
from implicit_image.models import Siren  # noqa: E402


def run(hidden, depth, size, steps=3, warm=1, fmt=0):
    eng = SirenEngine(size, size, hidden, depth, compute_dtype="f16", scratch_format=fmt)
    torch.manual_seed(0)
    init = Siren(depth=depth, hidden_size=hidden, first_omega_0=50.0, hidden_omega_0=30.0)
    dev = torch.device("cuda")
    eng.set_params(torch.cat([q.detach().reshape(-1) for q in init.parameters()]).to(dev))
    eng.set_coords(torch.linspace(0, 1, size).to(dev), torch.linspace(0, 1, size).to(dev))
    eng.set_target(torch.rand(size, size, 3, device=dev))
    eng.step([3e-4] * warm)
    torch.cuda.synchronize()
    eng.profile(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    eng.step([3e-4] * steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rep = eng.profile_report()
    pw = 2 * hidden + (depth - 2) * hidden * hidden + 3 * hidden
    F = 6 * pw - 4 * hidden
    out = {"hidden": hidden, "depth": depth, "size": size, "scratch_format": eng.scratch_format, "ms_per_step": dt * 1e3,
           "Mpix_iters_per_s": size * size / dt / 1e6, "step_TFLOPs": size * size * F / dt / 1e12,
           "kernels": {k: {"ms_per_step": v["total_ms"] / steps, "launches": v["launches"] / steps,
                           "tflops": v["flops_per_launch"] * v["launches"] / max(v["total_ms"], 1e-9) / 1e9}
                       for k, v in rep.items() if v["launches"]}}
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    fmts = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]     # e.g. "16,12": both phase formats
    for hidden, depth, size in ((512, 8, 2048), (1024, 12, 1024)):
        for fmt in fmts:
            run(hidden, depth, size, fmt=fmt)
