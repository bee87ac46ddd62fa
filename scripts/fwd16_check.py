"""Debug / evidence: k_fwd_pipe16 (16x16x32) against k_fwd_pipe (32x32x16) on the same weights: prediction, phase bytes,
dL/dout pieces and gradients.  Runs itself twice (the knob SIREN_FIT_FWD16 is read once per process)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))


def run(out):
    import torch
    from implicit_image._engine import SirenEngine
    from implicit_image.models import Siren
    H, W = 96, 80
    torch.manual_seed(0)
    init = torch.cat([q.detach().reshape(-1) for q in Siren(depth=8, hidden_size=256, first_omega_0=50., hidden_omega_0=30.).parameters()]).cuda()
    eng = SirenEngine(H, W, 256, 8, compute_dtype="f16", scratch_format=8)
    eng.set_params(init); eng.set_coords(torch.linspace(0, 1, H).cuda(), torch.linspace(0, 1, W).cuda())
    eng.set_target(torch.rand(H, W, 3, device="cuda"))
    pred, sse = eng.forward()
    loss = eng.forward_backward()
    np.savez(out, pred=pred.cpu().numpy(), sse=sse, loss=loss, phases=eng.debug_scratch("phases").cpu().numpy(),
             dlast=eng.debug_scratch("dlast").cpu().numpy(), grads=eng.get_grads().cpu().numpy())


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    outs = []
    for knob in ("0", "1"):
        o = f"/tmp/fwd16_{knob}.npz"
        subprocess.check_call([sys.executable, __file__, o], env=dict(os.environ, SIREN_FIT_FWD16=knob))
        outs.append(np.load(o))
    a, b = outs
    print("sse", float(a["sse"]), float(b["sse"]), "loss", float(a["loss"]), float(b["loss"]))
    print("pred max |d|", np.abs(a["pred"] - b["pred"]).max(), "first pixels", a["pred"].reshape(-1, 3)[:2], b["pred"].reshape(-1, 3)[:2])
    pa, pb = a["phases"].view(np.uint8), b["phases"].view(np.uint8)
    d = (pa.astype(np.int16) - pb.astype(np.int16) + 128) % 256 - 128
    n = 96 * 80 // 32 * 8 * 1024     # bytes of one layer plane that hold pixels
    print("phase bytes: planes", pa.size // max(n, 1), "differing", int((d != 0).sum()), "of", pa.size, "max |d|", int(np.abs(d).max()))
    for l in range(7):
        dl = d.reshape(-1)[l * (pa.size // 7):(l + 1) * (pa.size // 7)]
        print("  layer", l, "differing", int((dl != 0).sum()), "max", int(np.abs(dl).max()))
    da, db = a["dlast"].view(np.uint16), b["dlast"].view(np.uint16)
    print("dlast differing words", int((da != db).sum()), "of", da.size)
    ga, gb = a["grads"], b["grads"]
    print("grads rel", float(np.abs(ga - gb).max() / np.abs(ga).max()))
