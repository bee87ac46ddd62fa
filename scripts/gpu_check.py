"""Developer check: HIP engine vs CPU oracle on small cases (prints per-layer errors)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from oracle import siren_oracle as so
from implicit_image._engine import SirenEngine

def check(hidden, depth, H, W, dtype, chunk=0):
    torch.manual_seed(0)
    p = so.siren_init(hidden, depth, seed=0)
    img = so.synthetic_image(H, W, seed=7)
    grid = so.get_grid(H, W)
    loss, sse, grads = so.loss_and_grads(p, grid, img)
    pred_ref = so.forward(p, grid)
    eng = SirenEngine(H, W, hidden, depth, compute_dtype=dtype, chunk_pixels=chunk)
    gh, gw = so.grid_vectors(H, W)
    eng.set_coords(gh.cuda(), gw.cuda())
    eng.set_params(torch.tensor(so.flatten(p)).cuda())
    eng.set_target(img.cuda().contiguous())
    pred, sse_e = eng.forward()
    err = (pred.cpu() - pred_ref).abs().max().item()
    print(f"[{hidden}x{depth} {H}x{W} {dtype} chunk={chunk}] fwd max|dpred| {err:.3e}  sse {sse_e:.6f} vs {sse:.6f}")
    sse2 = eng.forward_backward()
    g = eng.get_grads().cpu().numpy()
    gref = so.flatten(grads)
    off = 0
    for l, (fin, fout) in enumerate(so.layer_dims(hidden, depth)):
        for nm, n in (("W", fin * fout), ("b", fout)):
            a, b = g[off:off + n], gref[off:off + n]; off += n
            print(f"   L{l}.{nm}: rel {np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30):.3e}  |ref| {np.linalg.norm(b):.3e} |eng| {np.linalg.norm(a):.3e}")
    print("   total grad rel err", np.linalg.norm(g - gref) / np.linalg.norm(gref), "sse(fb)", sse2)
    return eng

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    check(64, 4, 32, 40, "bf16")
    check(64, 4, 32, 40, "f16")
    check(256, 8, 32, 40, "f16")
    check(128, 6, 48, 56, "f16", chunk=1024)
    check(32, 3, 5, 7, "f16")
    # short training run: 100 steps 64x4 on 64x64 vs oracle
    H = W = 64
    p = so.siren_init(64, 4, seed=0); img = so.synthetic_image(H, W, seed=3); grid = so.get_grid(H, W)
    eng = SirenEngine(H, W, 64, 4, compute_dtype="f16")
    gh, gw = so.grid_vectors(H, W); eng.set_coords(gh.cuda(), gw.cuda())
    eng.set_params(torch.tensor(so.flatten(p)).cuda()); eng.set_target(img.cuda().contiguous())
    opt = so.Adam(p); lo = []
    for t in range(100): lo.append(so.train_epoch(p, opt, grid, img, t))
    le = eng.step([3e-4] * 100, want_loss=True)
    for t in (0, 1, 2, 10, 50, 99): print(f"   step {t}: oracle {lo[t]:.6f} engine {le[t]:.6f}")
    # timing at 1024x1024 256x8
    H = W = 1024
    eng = SirenEngine(H, W, 256, 8, compute_dtype="f16")
    gh, gw = so.grid_vectors(H, W); eng.set_coords(gh.cuda(), gw.cuda())
    eng.set_params(torch.tensor(so.flatten(so.siren_init(256, 8, seed=0))).cuda())
    img = torch.rand(H, W, 3, device="cuda"); eng.set_target(img)
    eng.step([3e-4] * 3); torch.cuda.synchronize()
    eng.profile(True); t0 = time.time(); eng.step([3e-4] * 5); torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"1024^2 256x8: {dt*1e3:.2f} ms/step -> {H*W/dt/1e6:.1f} Mpix-it/s")
    for k, v in eng.profile_report().items():
        if v["launches"]: print(f"   {k:12s} {v['total_ms']/5:8.3f} ms/step  launches/step {v['launches']/5:.0f}")
