"""Per-kernel ms of one fit step at SIZE^2 (default 2048) for the library selected by SIREN_FIT_LIB."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine, _LIB_PATH
from implicit_image.models import Siren
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
init = torch.cat([q.detach().reshape(-1) for q in Siren(depth=8, hidden_size=256, first_omega_0=50., hidden_omega_0=30.).parameters()]).cuda()
eng = SirenEngine(H, W, 256, 8, compute_dtype="f16")
eng.set_params(init); eng.set_coords(torch.linspace(0, 1, H).cuda(), torch.linspace(0, 1, W).cuda())
eng.set_target(torch.rand(H, W, 3, device="cuda"))
eng.step([3e-4] * 2); torch.cuda.synchronize()
eng.profile(True); eng.profile_reset()
t0 = time.perf_counter(); eng.step([3e-4] * 5); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(os.path.basename(_LIB_PATH), f"{dt*1e3:7.2f} ms/step {H*W/dt/1e6:6.1f} Mpix/s |", " ".join(f"{k[2:]}={v['total_ms']/5:.2f}" for k, v in eng.profile_report().items() if v['launches'] and v['total_ms']/5 > 0.05))
