#!/usr/bin/env python3
"""Golden vectors for the wide (hidden > 256) path, minted by running the REAL reference here
(same import recipe as make_golden.py; only data is written):

    python tests/golden/make_golden_wide.py

  wide_512x3_32.npz   seed-0 SIREN 512x3 on a 32x40 synthetic image: init, first-step loss / prediction /
                      dense gradients, and the 10-step Adam loss curve (train_epoch, lr 3e-4)
  sine_out_64x3_16.npz  the same quantities for SIREN 64x3 with outermost_linear=False (sine output layer) on 16x20
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402


def mint(th, siren, data, name, hidden, depth, H, W, outermost_linear):
    mlp = dict(first_omega_0=50.0, hidden_omega_0=30.0, outermost_linear=outermost_linear)
    img = mg.synthetic_image(H, W, seed=7)
    grid = data.get_grid(H, W)
    torch.manual_seed(0)
    m = siren.Siren(depth=depth, hidden_size=hidden, **mlp)
    p0 = np.concatenate([p.detach().numpy().ravel() for p in m.parameters()]).astype(np.float32)
    m.train()
    pred = m(grid)
    loss = torch.nn.functional.mse_loss(pred, img)
    loss.backward()
    grads = np.concatenate([p.grad.numpy().ravel() for p in m.parameters()]).astype(np.float32)
    m.zero_grad()
    optim, sched = th.get_optimizer_lr_scheduler(m, mg.Cfg(name="adam", lr=3e-4))
    ls = [th.train_epoch(m, optim, grid, img, lr_scheduler=sched) for _ in range(10)]
    np.savez_compressed(f"{mg.OUT}/{name}.npz", init=p0, loss=loss.item(), grads=grads,
                        pred=pred.detach().numpy(), img=img.numpy(), losses=np.array(ls, np.float64))
    print("%s: loss %.6f -> %.6f" % (name, ls[0], ls[-1]))


def main():
    th, siren, data, _ = mg.import_reference()
    mint(th, siren, data, "wide_512x3_32", 512, 3, 32, 40, True)
    mint(th, siren, data, "sine_out_64x3_16", 64, 3, 16, 20, False)


if __name__ == "__main__":
    main()
