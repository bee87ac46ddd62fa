"""Wide path: step time of SIREN 512x8 on 2048^2 (format 12) against the pixel chunk size - do smaller inter-kernel tensors
(closer to the 256 MB Infinity Cache) pay for their extra launches?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from implicit_image.models import Siren
hidden, depth, size = 512, 8, 2048
torch.manual_seed(0)
init = torch.cat([q.detach().reshape(-1) for q in Siren(depth=depth, hidden_size=hidden, first_omega_0=50.0, hidden_omega_0=30.0).parameters()]).cuda()
for chunk in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "131072,262144,524288,1048576,2097152").split(",")]:
    eng = SirenEngine(size, size, hidden, depth, compute_dtype="f16", scratch_format=12, chunk_pixels=chunk)
    eng.set_params(init); eng.set_coords(torch.linspace(0, 1, size).cuda(), torch.linspace(0, 1, size).cuda()); eng.set_target(torch.rand(size, size, 3, device="cuda"))
    eng.step([3e-4]); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.step([3e-4] * 3); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"chunk {chunk:8d} pixels: {dt*1e3:7.2f} ms/step  {size*size/dt/1e6:6.1f} Mpix/s")
    eng.close()
