"""Two forward_backward passes from the same state: per-layer bitwise equality of the gradients (race detector)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image._engine import SirenEngine
from oracle import siren_oracle as so
from bench import device_image
fmts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,8,16").split(",")]
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1024,2048,4096").split(",")]
hid, dep = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (256, 8)
for fmt in fmts:
    for H in sizes:
        p = so.siren_init(hid, dep, seed=0)
        eng = SirenEngine(H, H, hid, dep, compute_dtype="f16", scratch_format=fmt)
        eng.set_coords(torch.linspace(0, 1, H).cuda(), torch.linspace(0, 1, H).cuda())
        eng.set_params(torch.tensor(so.flatten(p)).cuda())
        eng.set_target(device_image(H, H, torch.device("cuda")))
        gs, hs = [], []
        for _ in range(3):
            eng.forward_backward(); gs.append(eng.get_grads().clone())
            torch.cuda.synchronize()
            ph = eng.debug_scratch("phases"); de = eng.debug_scratch("deltas"); dl = eng.debug_scratch("dlast"); sl = eng.debug_scratch("slabs")
            nl = dep - 1
            hs.append(([int(ph[i * (ph.numel() // nl):(i + 1) * (ph.numel() // nl)].to(torch.int64).sum()) for i in range(nl)],
                       [int(de[i * (de.numel() // nl):(i + 1) * (de.numel() // nl)].to(torch.int64).sum()) for i in range(nl)],
                       int(dl.to(torch.int64).sum())))
        print("   phase sums per layer equal:", ["=" if all(h[0][i] == hs[0][0][i] for h in hs) else "X" for i in range(dep - 1)],
              " delta sums:", ["=" if all(h[1][i] == hs[0][1][i] for h in hs) else "X" for i in range(dep - 1)],
              " dlast:", "=" if all(h[2] == hs[0][2] for h in hs) else "X")
        rows, off = [], 0
        for fin, fout in so.layer_dims(hid, dep):
            n = fin * fout + fout
            rows.append("".join("=" if torch.equal(gs[0][off:off+n], g[off:off+n]) else "X" for g in gs[1:]))
            off += n
        print(f"fmt {fmt} {H}^2 per-layer (run2,run3 vs run1): {' '.join(rows)}  nan={bool(torch.isnan(gs[0]).any())}", flush=True)
        eng.close()
