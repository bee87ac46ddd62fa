"""PSNR deviation from the real reference's config-1 fixture (64x4, 256x256, 1000 steps, un-annealed: chaotic to +-0.15 dB
under any perturbation) per scratch format; the annealed plateau fixtures are asserted in tests/test_gpu_parity.py."""
import math, os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "implicit-image-compression_amd"))
import test_gpu_parity as T
from oracle import siren_oracle as so
for fmt in (16, 12, 8):
    d = np.load(os.path.join(R, "tests/golden/hot_64x4_256.npz"))
    H = W = 256
    img = so.synthetic_image(H, W)
    eng = T._engine(H, W, 64, 4, "f16", so.unflatten(d["init"], 64, 4), img, scratch_format=fmt)
    eng.step([so.step_lr(3e-4, t) for t in range(1000)])
    _, sse = eng.forward(want_pred=False)
    c1 = 10 * math.log10(3 * H * W / sse) - float(d["psnr"])
    print(f"format {fmt}: config-1 dPSNR {c1:+.4f} dB")
