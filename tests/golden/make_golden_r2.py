#!/usr/bin/env python3
"""Round-2 golden vectors, minted by running the REAL reference in this container (same recipe as make_golden.py:
inert stubs for the absent non-arithmetic imports, models/siren.py loaded by path; only DATA is written).

    python tests/golden/make_golden_r2.py [plateau] [shapes] [rigl]

  plateau_256x8_{S}.npz   the metric model (SIREN 256x8) annealed to a plateau on the S x S formula image: 200 full-batch
                          steps of the reference's train_epoch with Adam lr 3e-4 and StepLR(40, 0.5) (a StepLR built
                          here and passed through train_epoch's own lr_scheduler argument, train_helper.py:183-184; the
                          reference's default StepLR(2000, .5) would leave a 200-step run mid-spike).  Run twice, with
                          8 and with 2 torch threads: `psnr_spread` is the reference's own summation-order noise.
  shapes_{W}x{D}.npz      BASELINE config 3 shapes (256x6, 512x6, 512x8) on a ragged 24x40 image: first-step loss,
                          prediction and dense gradients + a 10-step loss curve.
  rigl_256x8_48.npz       BASELINE config 4 at its real shape: SIREN 256x8, RigL density 0.1 (ERK), 100 steps on a
                          48x48 image with topology updates every 20 steps (i = 0, 20, 40, 60, 80: five updates, the
                          first at i = 0 as in compress.py:141-143), masks after every update, loss curve, PSNR
                          (8 and 2 threads -> spread), and the pre-update (w, grad, mask, rate) of every update so the
                          prune/grow decision can be replayed bit-exactly on the device.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import Cfg, flat_grads, flat_params, import_reference, synthetic_image  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
MLP = dict(name="siren", first_omega_0=50, hidden_omega_0=30, outermost_linear=True, simulate_quantization=False)


def plateau(th, siren, data, size, steps=200, lr_step=40):
    img = synthetic_image(size, size)
    grid = data.get_grid(size, size)
    out = {}
    for threads in (8, 2):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        m = siren.Siren(depth=8, hidden_size=256, **MLP)
        init = flat_params(m)
        optim, _ = th.get_optimizer_lr_scheduler(m, Cfg(name="adam", lr=3e-4))
        sched = torch.optim.lr_scheduler.StepLR(optim, lr_step, gamma=0.5)
        losses = [th.train_epoch(m, optim, grid, img, lr_scheduler=sched) for _ in range(steps)]
        _, l, psnr, psnr8 = th.eval_epoch(m, grid, img)
        out[threads] = (init, np.array(losses, np.float64), psnr, psnr8, flat_params(m))
        print(f"plateau_256x8_{size}: threads {threads} psnr {psnr:.4f} psnr8 {psnr8:.4f} loss[-1] {losses[-1]:.3e}", flush=True)
    torch.set_num_threads(8)
    a, b = out[8], out[2]
    np.savez_compressed(f"{OUT}/plateau_256x8_{size}.npz", init=a[0], losses=a[1], psnr=a[2], psnr8=a[3],
                        psnr_2threads=b[2], psnr_spread=abs(a[2] - b[2]), steps=steps, lr_step=lr_step,
                        final_head=a[4][:4096], size=size)


def shapes(th, siren, data):
    H, W = 24, 40
    img = synthetic_image(H, W, seed=21)
    grid = data.get_grid(H, W)
    for hidden, depth in ((256, 6), (512, 6), (512, 8)):
        torch.manual_seed(0)
        m = siren.Siren(depth=depth, hidden_size=hidden, **MLP)
        p0 = flat_params(m)
        m.train()
        pred = m(grid)
        loss = torch.nn.functional.mse_loss(pred, img)
        loss.backward()
        g0 = flat_grads(m)
        m.zero_grad()
        optim, sched = th.get_optimizer_lr_scheduler(m, Cfg(name="adam", lr=3e-4))
        ls = [th.train_epoch(m, optim, grid, img, lr_scheduler=sched) for _ in range(10)]
        np.savez_compressed(f"{OUT}/shapes_{hidden}x{depth}.npz", init=p0, loss=loss.item(), grads=g0,
                            pred=pred.detach().numpy(), img=img.numpy(), losses=np.array(ls, np.float64))
        print(f"shapes_{hidden}x{depth}: loss {loss.item():.6f} -> {ls[-1]:.6f}", flush=True)


def rigl(th, siren, data):
    hw, steps, interval, end_when = 48, 100, 20, 90
    img = synthetic_image(hw, hw, seed=13)
    grid = data.get_grid(hw, hw)
    res = {}
    for threads in (8, 2):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        m = siren.Siren(depth=8, hidden_size=256, **MLP)
        p0 = flat_params(m)
        optim, sched = th.get_optimizer_lr_scheduler(m, Cfg(name="adam", lr=3e-4))
        mcfg = Cfg(name="RigL", density=0.1, sparse_init="erdos-renyi-kernel", dense_gradients=True,
                   growth_mode="absolute-gradient", prune_mode="magnitude", redistribution_mode="none",
                   dense=False, prune_rate=0.1, decay_schedule="cosine", end_when=end_when, interval=interval)
        mask = th.setup_mask(m, optim, mcfg)
        names = list(mask.mask_dict.keys())

        def bits():
            return np.packbits(np.concatenate([mask.mask_dict[n].numpy().ravel().astype(np.uint8) for n in names]))
        rec = dict(init=p0, mask0=bits(), img=img.numpy())
        ls, dens, u = [], [], 0
        for i in range(steps):
            ls.append(th.train_epoch(m, optim, grid, img, lr_scheduler=sched, mask=mask))
            if i <= end_when and i % interval == 0:
                rec[f"upd{u}_w_in"] = flat_params(m)
                rec[f"upd{u}_g_in"] = flat_grads(m)
                rec[f"upd{u}_mask_in"] = bits()
                rec[f"upd{u}_rate"] = mask.prune_rate_decay.get_dr()
                mask.update_connections()
                rec[f"upd{u}_mask_out"] = bits()
                rec[f"upd{u}_w_out"] = flat_params(m)
                u += 1
            dens.append(mask.stats.total_density)
        _, l, psnr, psnr8 = th.eval_epoch(m, grid, img)
        rec.update(losses=np.array(ls, np.float64), density=np.array(dens), psnr=psnr, psnr8=psnr8, n_updates=u,
                   mask_step=mask.mask_step, nnz=np.array([int(mask.mask_dict[n].sum().item()) for n in names]))
        res[threads] = rec
        print(f"rigl_256x8_{hw}: threads {threads} psnr {psnr:.4f} density {dens[-1]:.4f} updates {u}", flush=True)
    torch.set_num_threads(8)
    a, b = res[8], res[2]
    a["psnr_2threads"] = b["psnr"]
    a["psnr_spread"] = abs(a["psnr"] - b["psnr"])
    a["masks_equal_2threads"] = bool(all(np.array_equal(a[f"upd{k}_mask_out"], b[f"upd{k}_mask_out"]) for k in range(a["n_updates"])))
    np.savez_compressed(f"{OUT}/rigl_256x8_{hw}.npz", **a)
    print("rigl spread", a["psnr_spread"], "masks equal across thread counts:", a["masks_equal_2threads"])


def main():
    what = sys.argv[1:] or ["plateau", "shapes", "rigl"]
    torch.set_num_threads(8)
    th, siren, data, _ = import_reference()
    if "shapes" in what:
        shapes(th, siren, data)
    if "rigl" in what:
        rigl(th, siren, data)
    if "plateau" in what:
        plateau(th, siren, data, int(os.environ.get("PLATEAU_SIZE", "256")))


if __name__ == "__main__":
    main()
