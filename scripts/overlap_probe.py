#!/usr/bin/env python3
"""Probe: do a forward-heavy and a backward-heavy phase overlap usefully on one GPU?  Two independent engines on
two streams step concurrently (their k_fwd / k_bwd phases drift against each other); compare the aggregate
Mpix-iters/s with one engine alone.  SIREN_FIT_BWD_WGS limits the persistent backward grids."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "implicit-image-compression_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from implicit_image._engine import SirenEngine  # noqa: E402
from implicit_image.models import Siren  # noqa: E402

torch.manual_seed(0)
INIT = torch.cat([q.detach().reshape(-1) for q in Siren(depth=8, hidden_size=256, first_omega_0=50., hidden_omega_0=30.).parameters()])

size, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2896, 4
dev = torch.device("cuda")


def make(stream):
    with torch.cuda.stream(stream):
        e = SirenEngine(size, size, 256, 8, compute_dtype="f16")
        g = torch.Generator(device=dev).manual_seed(0)
        e.set_params(INIT.to(dev))   # realistic weights: random small weights overflow the backward and run at a higher clock
        e.set_coords(torch.linspace(0, 1, size).to(dev), torch.linspace(0, 1, size).to(dev))
        e.img = torch.rand(size, size, 3, device=dev, generator=g)
        e.set_target(e.img)
    stream.synchronize()
    return e


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
e1, e2 = make(s1), make(s2)
e1.step([3e-4]); e2.step([3e-4]); torch.cuda.synchronize()
t0 = time.perf_counter(); e1.step([3e-4] * steps); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(f"one engine : {size * size * steps / t1 / 1e6:.1f} Mpix-it/s")
ths = [threading.Thread(target=lambda e=e: e.step([3e-4] * steps)) for e in (e1, e2)]
t0 = time.perf_counter()
[t.start() for t in ths]; [t.join() for t in ths]; torch.cuda.synchronize()
t2 = time.perf_counter() - t0
print(f"two engines: {2 * size * size * steps / t2 / 1e6:.1f} Mpix-it/s aggregate")
