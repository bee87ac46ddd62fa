"""Tiny driver for rocprofv3 counter passes: a few fit steps of SIREN 256x8 at SIZE^2 (default 2048)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from implicit_image.models import Siren
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(0)
init = torch.cat([q.detach().reshape(-1) for q in Siren(depth=8, hidden_size=256, first_omega_0=50., hidden_omega_0=30.).parameters()]).cuda()
eng = SirenEngine(H, W, 256, 8, compute_dtype="f16")
eng.set_params(init); eng.set_coords(torch.linspace(0, 1, H).cuda(), torch.linspace(0, 1, W).cuda())
eng.set_target(torch.rand(H, W, 3, device="cuda"))
eng.step([3e-4] * steps); torch.cuda.synchronize()
print("done")
