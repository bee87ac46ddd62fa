"""Child of tests/test_gpu_fwd_tilings.py: one forward + forward_backward + 20 steps of a SIREN (256x8 unless given) on a ragged
image, with the forward kernel chosen by an environment knob the library reads once per process (SIREN_FIT_FWD16,
SIREN_FIT_WGEMM3).  Writes an npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "implicit-image-compression_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    out, H, W, fmt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    hidden, depth = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (256, 8)
    from implicit_image._engine import SirenEngine
    from oracle import siren_oracle as so           # (test infrastructure: the seed-0 init and the image formula)
    p = so.siren_init(hidden, depth, seed=0)
    eng = SirenEngine(H, W, hidden, depth, compute_dtype="f16", scratch_format=fmt)
    gh, gw = so.grid_vectors(H, W)
    eng.set_coords(gh.cuda(), gw.cuda())
    eng.set_params(torch.tensor(so.flatten(p)).cuda())
    eng.set_target(so.synthetic_image(H, W, seed=5).cuda().contiguous())
    pred, sse = eng.forward()
    loss = eng.forward_backward()
    grads = eng.get_grads().cpu().numpy()
    phases, dlast = eng.debug_scratch("phases").cpu().numpy(), eng.debug_scratch("dlast").cpu().numpy()   # (of that pass)
    losses = eng.step([3e-4] * 20, want_loss=True)
    np.savez(out, pred=pred.cpu().numpy(), sse=sse, loss=loss, phases=phases, dlast=dlast, grads=grads, losses=np.array(losses))


if __name__ == "__main__":
    main()
