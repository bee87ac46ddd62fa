"""8-bit vs 16-bit scratch on large grids: gradient agreement (relative L2, per layer) and loss curves."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from oracle import siren_oracle as so
sys.path.insert(0, ROOT)
from bench import device_image

def mk(H, W, scratch, chunk=0):
    p = so.siren_init(256, 8, seed=0)
    eng = SirenEngine(H, W, 256, 8, compute_dtype="f16", scratch_format=scratch, chunk_pixels=chunk)
    eng.set_coords(torch.linspace(0, 1, H).cuda(), torch.linspace(0, 1, W).cuda())
    eng.set_params(torch.tensor(so.flatten(p)).cuda())
    return eng

for (H, chunk) in [(1024, 0), (2048, 0), (2048, 1 << 20), (4096, 0)]:
    img = device_image(H, H, torch.device("cuda"))
    a, b = mk(H, H, 16, chunk), mk(H, H, 8, chunk)
    a.set_target(img); b.set_target(img)
    sa, sb = a.forward_backward(), b.forward_backward()
    ga, gb = a.get_grads(), b.get_grads()
    per, off = [], 0
    for fin, fout in so.layer_dims(256, 8):
        n = fin * fout
        per.append("%.1e" % ((ga[off:off+n] - gb[off:off+n]).norm() / ga[off:off+n].norm()).item())
        off += n + fout
    print(f"{H}^2 chunk {chunk}: sse {sa:.6e} {sb:.6e} grad rel L2 {((ga-gb).norm()/ga.norm()).item():.2e} per-layer {per}", flush=True)
    la = a.step([3e-4] * 25, want_loss=True); lb = b.step([3e-4] * 25, want_loss=True)
    print("   loss16", ["%.5f" % x for x in la[::4]]); print("   loss8 ", ["%.5f" % x for x in lb[::4]], flush=True)
    del a, b
