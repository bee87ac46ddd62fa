"""Wide path (hidden 512 / 1024) with fp8 deltas (scratch_format 8) against formats 12 / 16 and the fp32 oracle: per-tensor gradient
error on small grids, PSNR after 60 steps (the wide parity case), and - optionally - the bench shapes."""
import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from oracle import siren_oracle as so
def mk(H, W, hidden, depth, p, img, fmt, chunk=0):
    e = SirenEngine(H, W, hidden, depth, compute_dtype="f16", scratch_format=fmt, chunk_pixels=chunk)
    gh, gw = so.grid_vectors(H, W)
    e.set_coords(gh.cuda(), gw.cuda()); e.set_params(torch.tensor(so.flatten(p)).cuda()); e.set_target(img.cuda().contiguous())
    return e
for (H, W, hidden, depth, chunk) in [(40, 52, 512, 5, 0), (9, 33, 1024, 3, 0), (48, 48, 1024, 4, 1024), (24, 40, 512, 8, 0), (128, 128, 512, 6, 0)]:
    p = so.siren_init(hidden, depth, seed=3); img = so.synthetic_image(H, W, seed=5); grid = so.get_grid(H, W)
    loss, sse_ref, grads = so.loss_and_grads(p, grid, img); ref = so.flatten(grads)
    for fmt in (12, 8):
        e = mk(H, W, hidden, depth, p, img, fmt, chunk)
        e.forward_backward(); g = e.get_grads().cpu().numpy()
        errs = []; off = 0
        for fin, fout in so.layer_dims(hidden, depth):
            for n in (fin * fout, fout):
                errs.append(np.linalg.norm(g[off:off+n] - ref[off:off+n]) / (np.linalg.norm(ref[off:off+n]) + 1e-30)); off += n
        print(f"{H}x{W} {hidden}x{depth} fmt {fmt} ({e.scratch_format}): total rel {np.linalg.norm(g-ref)/np.linalg.norm(ref):.2e} per W tensor " + " ".join("%.1e" % x for x in errs[::2]), "finite", bool(np.isfinite(g).all()))
        e.close()
H = W = 96; hidden, depth, steps = 512, 4, 60
img, grid = so.synthetic_image(H, W, seed=8), so.get_grid(H, W)
for fmt in (12, 8):
    p = so.siren_init(hidden, depth, seed=0)
    e = mk(H, W, hidden, depth, p, img, fmt)
    e.step([so.step_lr(3e-4, t) for t in range(steps)])
    _, sse = e.forward(want_pred=False)
    print("psnr after 60 steps fmt", fmt, 10 * math.log10(3 * H * W / sse))
p = so.siren_init(hidden, depth, seed=0); opt = so.Adam(p)
for t in range(steps): so.train_epoch(p, opt, grid, img, t)
print("oracle", so.eval_epoch(p, grid, img)[2])
