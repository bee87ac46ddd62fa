"""CPU: the 200-step annealed fit of the non-smooth plateau fixture with (a) the fp32 oracle and (b) the engine's numerics
model (oracle/engine_model.py) per scratch format: which part of the engine's PSNR deviation on non-smooth content is
operand rounding and which is something else?   usage: python scripts/ns_model_probe.py [size] [formats e.g. 16,12,0; -1,-2,.. = fp32 + 1e-3 gradient noise, seed 1,2,..]"""
import math, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import siren_oracle as so, engine_model as em
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fmts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,16,12").split(",")]
steps, lr_step = 200, 40
img, grid = so.nonsmooth_image(S, S), so.get_grid(S, S)
torch.set_num_threads(int(os.environ.get("THREADS", "4")))
for fmt in fmts:
    p = so.siren_init(256, 8, seed=0)
    opt = so.Adam(p)
    t0 = time.time()
    for t in range(steps):
        if fmt <= 0:
            loss, _, grads = so.loss_and_grads(p, grid, img)
            if fmt < 0:   # CONTROL: the fp32 reference arithmetic with 1e-3 relative Gaussian noise on every gradient tensor (seed -fmt)
                gen = torch.Generator().manual_seed(1000 * (-fmt) + t)
                grads = [g + 1e-3 * g.norm() / math.sqrt(g.numel()) * torch.randn(g.shape, generator=gen) for g in grads]
        else:
            loss, _, grads, _ = em.loss_and_grads(p, grid, img, scratch=fmt)
        opt.step(p, grads, lr=3e-4 * 0.5 ** (t // lr_step))
        if t % 20 == 0: print(f"  fmt {fmt} step {t} loss {loss:.6e} ({time.time()-t0:.0f}s)", flush=True)
    mse, psnr, _ = so.metrics(so.forward(p, grid), img)
    print(f"fmt {fmt}: PSNR {psnr:.4f} (fp32 forward of the final weights)", flush=True)
    # results accumulate in a small fixture the GPU test reads (tests/golden/plateau_ns_256x8_{S}_model.npz)
    out = os.path.join(ROOT, "tests", "golden", f"plateau_ns_256x8_{S}_model.npz")
    rec = dict(np.load(out)) if os.path.exists(out) else {}
    rec[f"psnr_fmt{fmt}" if fmt >= 0 else f"psnr_noise_seed{-fmt}"] = np.float64(psnr)
    np.savez(out, **rec)
