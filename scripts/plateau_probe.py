"""256x8 annealed plateau (tests/golden/plateau_256x8_256.npz): engine PSNR per scratch format / dtype vs the reference."""
import math, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from oracle import siren_oracle as so
d = np.load(os.path.join(ROOT, "tests/golden/plateau_256x8_256.npz"))
S, steps, lr_step = int(d["size"]), int(d["steps"]), int(d["lr_step"])
img = so.synthetic_image(S, S)
p = so.siren_init(256, 8, seed=0)
lrs = [3e-4 * 0.5 ** (t // lr_step) for t in range(steps)]
print("reference psnr", float(d["psnr"]), "spread", float(d["psnr_spread"]))
for dtype, fmt in (("f16", 16), ("f16", 12), ("f16", 8), ("bf16", 16)):
    eng = SirenEngine(S, S, 256, 8, compute_dtype=dtype, scratch_format=fmt)
    gh, gw = so.grid_vectors(S, S)
    eng.set_coords(gh.cuda(), gw.cuda()); eng.set_params(torch.tensor(so.flatten(p)).cuda()); eng.set_target(img.cuda().contiguous())
    losses = eng.step(lrs, want_loss=True)
    _, sse = eng.forward(want_pred=False)
    psnr = 10 * math.log10(3 * S * S / sse)
    ref = d["losses"]
    print(f"{dtype} fmt {fmt}: psnr {psnr:.4f}  d {psnr - float(d['psnr']):+.4f}  loss rel err @10/50/199: "
          f"{abs(losses[10]/ref[10]-1):.2e} {abs(losses[50]/ref[50]-1):.2e} {abs(losses[199]/ref[199]-1):.2e}", flush=True)
