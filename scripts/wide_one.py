#!/usr/bin/env python3
"""Two steps of SIREN 1024x12 on 1024x1024 (one chunk) for rocprofv3 counter passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "implicit-image-compression_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from implicit_image._engine import SirenEngine  # noqa: E402

hidden, depth, size = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 12, 1024)))
eng = SirenEngine(size, size, hidden, depth, compute_dtype="f16")
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
eng.set_params((torch.rand(eng.num_params, device=dev, generator=g) * 2 - 1) * 0.03)
eng.set_coords(torch.linspace(0, 1, size).to(dev), torch.linspace(0, 1, size).to(dev))
eng.set_target(torch.rand(size, size, 3, device=dev, generator=g))
eng.step([3e-4] * 2)
torch.cuda.synchronize()
eng.close()
