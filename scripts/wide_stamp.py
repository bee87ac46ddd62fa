"""Runs a few steps of SIREN 512x8 on 2048^2 (format 12) with the library in SIREN_FIT_LIB and closes the engine: an
SF_WEXP_STAMP build prints k_wgemm2's cycles per tile (all / waits at the chunk barriers / epilogue) at sf_destroy."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from implicit_image.models import Siren
hidden, depth, size = 512, 8, 2048
torch.manual_seed(0)
init = torch.cat([q.detach().reshape(-1) for q in Siren(depth=depth, hidden_size=hidden, first_omega_0=50.0, hidden_omega_0=30.0).parameters()]).cuda()
eng = SirenEngine(size, size, hidden, depth, compute_dtype="f16", scratch_format=12)
eng.set_params(init); eng.set_coords(torch.linspace(0, 1, size).cuda(), torch.linspace(0, 1, size).cuda()); eng.set_target(torch.rand(size, size, 3, device="cuda"))
eng.step([3e-4]); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.step([3e-4] * 3); torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / 3 * 1e3:.2f} ms/step", flush=True)
eng.close()
