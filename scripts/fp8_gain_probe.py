"""Does the weight norm predict the layer-to-layer growth of the hidden deltas?  gain_l = omega * sqrt(0.5 * ||W_l||_F^2 / n_in)
(independent delta components) against the measured rms ratio of the fp8 deltas, during a fit of the non-smooth image."""
import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from oracle import siren_oracle as so
kind = sys.argv[1] if len(sys.argv) > 1 else "nonsmooth"
H = W = 256
img = so.nonsmooth_image(H, W) if kind == "nonsmooth" else so.synthetic_image(H, W, seed=5)
p = so.siren_init(256, 8, seed=0)
eng = SirenEngine(H, W, 256, 8, compute_dtype="f16", scratch_format=8)
gh, gw = so.grid_vectors(H, W)
eng.set_coords(gh.cuda(), gw.cuda()); eng.set_params(torch.tensor(so.flatten(p)).cuda()); eng.set_target(img.cuda().contiguous())
lut = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().cuda()
done = 0
for mk in (0, 50, 100, 200, 400, 1000):
    if mk > done:
        eng.step([3e-4 * 0.5 ** (t // 200) for t in range(done, mk)]); done = mk
    eng.forward_backward()
    d = eng.debug_scratch("deltas"); n = d.numel() // 7
    rms = [float(lut[d[l * n:l * n + H * W * 256].long()].pow(2).mean().sqrt()) for l in range(7)]
    flat = eng.get_params().cpu().numpy()
    pr = so.unflatten(flat, 256, 8)
    pred = []
    for l in range(1, 7):     # W_l maps layer l-1 -> l; delta_{l-1} from delta_l
        Wl = np.asarray(pr[2 * l], dtype=np.float64)
        pred.append(30.0 * math.sqrt(0.5 * (Wl ** 2).sum() / Wl.shape[1]))
    meas = [rms[l - 1] / rms[l] for l in range(1, 7)]
    Wo = np.asarray(pr[14], dtype=np.float64)
    print("step %4d  measured gain L(l)->L(l-1), l=1..6: %s | predicted: %s | cumulative measured %.2f predicted %.2f | rms L6 %.3g, ||W_out||_F %.3g" % (
        mk, " ".join("%.2f" % m for m in meas), " ".join("%.2f" % q for q in pred), rms[0] / rms[6], float(np.prod(pred)), rms[6], math.sqrt((Wo ** 2).sum())))
