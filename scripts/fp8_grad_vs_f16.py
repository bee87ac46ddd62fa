"""Per-layer accuracy of the format-8 gradient against the format-16 one on the same weights, over the first steps of the non-smooth fit\n(evidence that the per-layer fp8 scales leave the early gradients as accurate as the single scale did: <= 1.3 % per layer)."""
import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from oracle import siren_oracle as so
H = W = 256
img = so.nonsmooth_image(H, W); p = so.siren_init(256, 8, seed=0)
gh, gw = so.grid_vectors(H, W)
def mk(fmt):
    e = SirenEngine(H, W, 256, 8, compute_dtype="f16", scratch_format=fmt)
    e.set_coords(gh.cuda(), gw.cuda()); e.set_params(torch.tensor(so.flatten(p)).cuda()); e.set_target(img.cuda().contiguous())
    return e
e16, e8 = mk(16), mk(8)
for it in range(0, 12):
    e16.forward_backward(); e8.set_params(e16.get_params()); e8.forward_backward()
    g16, g8 = e16.get_grads().cpu().numpy(), e8.get_grads().cpu().numpy()
    out = []
    sizes = [512] + [65536] * 6 + [768]
    for l in range(8):
        ow = e16.param_offsets(l)[0]
        a, b = g16[ow:ow + sizes[l]], g8[ow:ow + sizes[l]]
        out.append("%.3f/%.3f" % (np.linalg.norm(b - a) / np.linalg.norm(a), np.linalg.norm(b) / np.linalg.norm(a)))
    print("step", it, "per layer W: rel |g8 - g16| / |g16| and |g8| / |g16|:", " ".join(out))
    e16.adam_step(3e-4)
