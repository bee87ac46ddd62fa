"""Per-layer statistics of the fp8 (e4m3) hidden deltas of a format-8 fit: rms in fp8 units, share of saturated (+-448), zero
and subnormal (< 2^-6) bytes - evidence for the choice of the per-chunk scale target (SIREN_FIT_FP8_TARGET).
usage: python scripts/fp8_stats.py [smooth|nonsmooth] [steps,steps,...]"""
import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from oracle import siren_oracle as so

kind = sys.argv[1] if len(sys.argv) > 1 else "nonsmooth"
marks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,100,400,1000").split(",")]
H = W = 256
img = so.nonsmooth_image(H, W) if kind == "nonsmooth" else so.synthetic_image(H, W, seed=5)
p = so.siren_init(256, 8, seed=0)
eng = SirenEngine(H, W, 256, 8, compute_dtype="f16", scratch_format=8)
gh, gw = so.grid_vectors(H, W)
eng.set_coords(gh.cuda(), gw.cuda()); eng.set_params(torch.tensor(so.flatten(p)).cuda()); eng.set_target(img.cuda().contiguous())
lut = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().cuda()
done = 0
print("target", os.environ.get("SIREN_FIT_FP8_TARGET", "default"), kind)
for mk in marks:
    if mk > done:
        eng.step([3e-4 * 0.5 ** (t // 200) for t in range(done, mk)]); done = mk
    loss = eng.forward_backward()
    d = eng.debug_scratch("deltas")
    n = d.numel() // 7
    line = []
    for l in range(7):
        b = d[l * n:l * n + H * W * 256]
        v = lut[b.long()]
        a = v.abs()
        line.append("L%d rms %.3g sat %.2e zero %.3f sub %.3f" % (l, float(v.pow(2).mean().sqrt()), float((a >= 448).float().mean()), float((a == 0).float().mean()), float(((a > 0) & (a < 2 ** -6)).float().mean())))
    print("step %4d loss %.3e | " % (mk, loss / (3 * H * W)) + " | ".join(line))
