import os, sys, hashlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from oracle import siren_oracle as so
H = W = 256
img = so.nonsmooth_image(H, W); p = so.siren_init(256, 8, seed=0)
eng = SirenEngine(H, W, 256, 8, compute_dtype="f16", scratch_format=int(sys.argv[1]))
gh, gw = so.grid_vectors(H, W)
eng.set_coords(gh.cuda(), gw.cuda()); eng.set_params(torch.tensor(so.flatten(p)).cuda()); eng.set_target(img.cuda().contiguous())
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
l = eng.forward_backward(); g = eng.get_grads().cpu().numpy()
print("fb loss", l, "grads", sha(g), "phases", sha(eng.debug_scratch("phases").cpu().numpy()), "deltas", sha(eng.debug_scratch("deltas").cpu().numpy()))
for i in range(8):
    off, n = eng.param_offsets(i)[0], 0
ls = eng.step([3e-4] * 3, want_loss=True)
print("losses", ls, "params", sha(eng.get_params().cpu().numpy()))
