"""Per-kernel ms of forward_backward ONLY (no Adam: the weights stay the SIREN init, so timing-only builds whose gradients
are garbage do not turn the operands into NaN and change the power state) at SIZE^2 (default 4096), library from
SIREN_FIT_LIB.  usage: python scripts/fb_bench.py [SIZE] [REPS]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine, _LIB_PATH
from implicit_image.models import Siren
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fmt = int(os.environ.get("FMT", "0"))
torch.manual_seed(0)
init = torch.cat([q.detach().reshape(-1) for q in Siren(depth=8, hidden_size=256, first_omega_0=50., hidden_omega_0=30.).parameters()]).cuda()
eng = SirenEngine(H, W, 256, 8, compute_dtype="f16", scratch_format=fmt)
eng.set_params(init); eng.set_coords(torch.linspace(0, 1, H).cuda(), torch.linspace(0, 1, W).cuda())
eng.set_target(torch.rand(H, W, 3, device="cuda"))
for _ in range(3): eng.forward_backward()
torch.cuda.synchronize()
eng.profile(True); eng.profile_reset()
t0 = time.perf_counter()
for _ in range(reps): eng.forward_backward(sync=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(os.path.basename(_LIB_PATH), f"fmt {fmt}: {dt*1e3:7.2f} ms/pass |", " ".join(f"{k[2:]}={v['total_ms']/reps:.2f}" for k, v in eng.profile_report().items() if v['launches'] and v['total_ms']/reps > 0.05))
