#!/usr/bin/env python3
"""Mint golden vectors by running the REAL reference in this container.

Run once, here (the reference cannot travel to the GPU box):

    python tests/golden/make_golden.py

It imports the reference's own `train_epoch` / `eval_epoch` / `setup_mask` /
`Siren` / `Masking` from /root/reference (read-only) exactly as SURVEY.md §8(c)
describes: inert `sys.modules` stubs for the absent, non-arithmetic imports
(`omegaconf`, `torch_optimizer`, and `cv2` / `kornia` / `matplotlib` for
`data.get_grid`), `models/siren.py` loaded by file path because the package
`__init__` eagerly imports `wavelet_siren`.  Only DATA (inputs and expected
outputs) is written under tests/golden/; no reference source is copied.

Fixtures written (all small):
  hot_64x4_256.npz     config 1: seed-0 init, 1000-step loss curve, final PSNRs, final weights
  grads_*.npz          first-step loss + dense gradients on a small image (64x4, 256x8, 128x6)
  short_256x8_64.npz   20-step loss curve at 256x8 on a 64x64 image
  grid.npz             get_grid(5,7) and the 4096-point linspace vector
  erk_masks.npz        ERK masks (packed bits) + nnz tables
  truncate_*.npz       one truncate_weights() in/out pair (w, grad, mask, rate -> mask', w')
  cosine_decay.npz     CosineDecay sequence incl. the mask_step double increment
  rigl_64x4_64.npz     120-step RigL run: loss curve, masks, prune-rate/density trace
"""
import hashlib
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    sys.path.insert(0, REF)
    _stub("omegaconf", DictConfig=dict, OmegaConf=object)
    _stub("torch_optimizer", Shampoo=object)
    _stub("cv2")
    _stub("kornia")
    mpl = _stub("matplotlib")
    mpl.pyplot = _stub("matplotlib.pyplot")
    from implicit_image.utils import train_helper  # noqa
    from implicit_image.pipeline.masking.funcs import decay  # noqa
    spec = importlib.util.spec_from_file_location("ref_siren", f"{REF}/implicit_image/models/siren.py")
    siren = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(siren)
    spec = importlib.util.spec_from_file_location("ref_data", f"{REF}/implicit_image/data.py")
    data = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(data)
    return train_helper, siren, data, decay


class Cfg(dict):
    __getattr__ = dict.get


def synthetic_image(H, W, seed=1234):
    """SURVEY §8(d) formula image: smooth sinusoids + seeded uniform noise (amp 0.05)."""
    ys = torch.linspace(0, 1, H)[:, None].expand(H, W)
    xs = torch.linspace(0, 1, W)[None, :].expand(H, W)
    kx = torch.tensor([1.0, 2.0, 3.0])
    ky = torch.tensor([3.0, 1.0, 2.0])
    img = 0.5 + 0.25 * torch.sin(12 * xs[..., None] * kx) + 0.25 * torch.cos(9 * ys[..., None] * ky)
    g = torch.Generator().manual_seed(seed)
    img = img + 0.05 * (torch.rand(H, W, 3, generator=g) * 2 - 1)
    return img.clamp(0, 1).float().contiguous()


def flat_params(model):
    return np.concatenate([p.detach().numpy().ravel() for p in model.parameters()]).astype(np.float32)


def flat_grads(model):
    return np.concatenate([p.grad.detach().numpy().ravel() for p in model.parameters()]).astype(np.float32)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    torch.set_num_threads(8)
    th, siren, data, decay = import_reference()
    mlp = dict(name="siren", first_omega_0=50, hidden_omega_0=30, outermost_linear=True,
               simulate_quantization=False)

    # ---- grid (data.py:78-88) -------------------------------------------------
    g57 = data.get_grid(5, 7).numpy()
    np.savez_compressed(f"{OUT}/grid.npz", grid_5x7=g57,
                        lin4096=torch.linspace(0, 1, 4096).numpy(),
                        lin256=torch.linspace(0, 1, 256).numpy(),
                        lin7=torch.linspace(0, 1, 7).numpy())

    # ---- config 1: 64x4 on 256x256, 1000 dense steps ---------------------------
    H = W = 256
    img = synthetic_image(H, W)
    grid = data.get_grid(H, W)
    torch.manual_seed(0)
    model = siren.Siren(depth=4, hidden_size=64, **mlp)
    init = flat_params(model)
    optim, sched = th.get_optimizer_lr_scheduler(model, Cfg(name="adam", lr=3e-4))
    losses = []
    for i in range(1000):
        losses.append(th.train_epoch(model, optim, grid, img, lr_scheduler=sched))
    pred, l, psnr, psnr8 = th.eval_epoch(model, grid, img)
    np.savez_compressed(f"{OUT}/hot_64x4_256.npz", init=init, losses=np.array(losses, np.float64),
                        final=flat_params(model), eval_loss=l, psnr=psnr, psnr8=psnr8,
                        img_sha256=sha(img.numpy()), img_corner=img[:4, :4].numpy(),
                        pred_corner=pred[:4, :4].numpy())
    print("hot_64x4_256: loss0 %.6f lossN %.6f psnr %.4f psnr8 %.4f" % (losses[0], losses[-1], psnr, psnr8))

    # ---- first-step gradients on a small image ---------------------------------
    for hidden, depth, hw in ((64, 4, 32), (256, 8, 32), (128, 6, 48)):
        img_s = synthetic_image(hw, hw + 8, seed=7)   # non-square on purpose
        grid_s = data.get_grid(hw, hw + 8)
        torch.manual_seed(0)
        m = siren.Siren(depth=depth, hidden_size=hidden, **mlp)
        p0 = flat_params(m)
        m.train()
        pred = m(grid_s)
        loss = torch.nn.functional.mse_loss(pred, img_s)
        loss.backward()
        np.savez_compressed(f"{OUT}/grads_{hidden}x{depth}_{hw}.npz", init=p0, loss=loss.item(),
                            grads=flat_grads(m), pred=pred.detach().numpy(), img=img_s.numpy())
        print(f"grads_{hidden}x{depth}_{hw}: loss {loss.item():.6f}")

    # ---- 20 steps at 256x8 on 64x64 --------------------------------------------
    img_s = synthetic_image(64, 64, seed=11)
    grid_s = data.get_grid(64, 64)
    torch.manual_seed(0)
    m = siren.Siren(depth=8, hidden_size=256, **mlp)
    p0 = flat_params(m)
    optim, sched = th.get_optimizer_lr_scheduler(m, Cfg(name="adam", lr=3e-4))
    ls = [th.train_epoch(m, optim, grid_s, img_s, lr_scheduler=sched) for _ in range(20)]
    _, l, psnr, psnr8 = th.eval_epoch(m, grid_s, img_s)
    np.savez_compressed(f"{OUT}/short_256x8_64.npz", init=p0, losses=np.array(ls, np.float64),
                        final=flat_params(m), psnr=psnr, psnr8=psnr8, img=img_s.numpy())
    print("short_256x8_64: psnr %.4f" % psnr)

    # ---- ERK masks (core.py:220-248, init_scheme.py:40-158) ----------------------
    erk = {}
    for hidden, depth, density in ((256, 8, 0.1), (64, 4, 0.5), (128, 8, 0.5)):
        torch.manual_seed(0)
        m = siren.Siren(depth=depth, hidden_size=hidden, **mlp)
        optim, _ = th.get_optimizer_lr_scheduler(m, Cfg(name="adam", lr=3e-4))
        mcfg = Cfg(name="RigL", density=density, sparse_init="erdos-renyi-kernel", dense_gradients=True,
                   growth_mode="absolute-gradient", prune_mode="magnitude", redistribution_mode="none",
                   dense=False, prune_rate=0.1, decay_schedule="cosine", end_when=1500, interval=20)
        mask = th.setup_mask(m, optim, mcfg)
        key = f"{hidden}x{depth}_d{density}"
        names = list(mask.mask_dict.keys())
        erk[key + "_nnz"] = np.array([int(mask.mask_dict[n].sum().item()) for n in names], np.int64)
        erk[key + "_bits"] = np.packbits(np.concatenate(
            [mask.mask_dict[n].numpy().ravel().astype(np.uint8) for n in names]))
        erk[key + "_baseline_nonzero"] = mask.baseline_nonzero
        erk[key + "_total_params"] = mask.total_params
        erk[key + "_rng_after"] = torch.rand(4).numpy()   # pins the generator position after add_module
        erk[key + "_params_after"] = flat_params(m)       # weights after apply_mask
        print("erk", key, erk[key + "_nnz"].tolist(), mask.baseline_nonzero, mask.total_params)
    np.savez_compressed(f"{OUT}/erk_masks.npz", **erk)

    # ---- one truncate_weights() in/out pair ---------------------------------------
    for hidden, depth, density, hw in ((64, 4, 0.5, 32), (256, 8, 0.1, 16)):
        torch.manual_seed(0)
        m = siren.Siren(depth=depth, hidden_size=hidden, **mlp)
        optim, sched = th.get_optimizer_lr_scheduler(m, Cfg(name="adam", lr=3e-4))
        mcfg = Cfg(name="RigL", density=density, sparse_init="erdos-renyi-kernel", dense_gradients=True,
                   growth_mode="absolute-gradient", prune_mode="magnitude", redistribution_mode="none",
                   dense=False, prune_rate=0.1, decay_schedule="cosine", end_when=1500, interval=20)
        mask = th.setup_mask(m, optim, mcfg)
        img_s = synthetic_image(hw, hw, seed=5)
        grid_s = data.get_grid(hw, hw)
        for _ in range(3):
            th.train_epoch(m, optim, grid_s, img_s, lr_scheduler=sched, mask=mask)
        names = list(mask.mask_dict.keys())
        w_in = flat_params(m)
        g_in = flat_grads(m)
        mask_in = np.packbits(np.concatenate([mask.mask_dict[n].numpy().ravel().astype(np.uint8) for n in names]))
        rate_in = mask.prune_rate_decay.get_dr()
        step_in = mask.mask_step
        mask.update_connections()
        mask_out = np.packbits(np.concatenate([mask.mask_dict[n].numpy().ravel().astype(np.uint8) for n in names]))
        np.savez_compressed(f"{OUT}/truncate_{hidden}x{depth}.npz", w_in=w_in, g_in=g_in, mask_in=mask_in,
                            rate_in=rate_in, mask_step_in=step_in, mask_out=mask_out, w_out=flat_params(m),
                            mask_step_out=mask.mask_step,
                            removed=np.array([mask.stats.removed_dict.get(n, 0) for n in names]),
                            nnz_out=np.array([int(mask.mask_dict[n].sum().item()) for n in names]))
        print("truncate", hidden, depth, "rate", rate_in, "steps", step_in, mask.mask_step)

    # ---- CosineDecay sequence with the reference's stepping pattern ------------------
    # (core.py:696-702 step(mask_step) then +=1 ; core.py:771 another += 1 on update)
    d = decay.CosineDecay(prune_rate=0.1, T_max=60)
    seq, mask_step = [], 0
    for i in range(100):
        d.step(mask_step)
        mask_step += 1
        if i <= 60 and i % 5 == 0:
            mask_step += 1
        seq.append(d.get_dr())
    np.savez_compressed(f"{OUT}/cosine_decay.npz", seq=np.array(seq, np.float64), T_max=60, interval=5,
                        prune_rate=0.1)

    # ---- RigL end-to-end, 64x4 on 64x64, 120 steps, interval 10, end_when 90 ---------
    hw = 64
    img_s = synthetic_image(hw, hw, seed=3)
    grid_s = data.get_grid(hw, hw)
    torch.manual_seed(0)
    m = siren.Siren(depth=4, hidden_size=64, **mlp)
    p0 = flat_params(m)
    optim, sched = th.get_optimizer_lr_scheduler(m, Cfg(name="adam", lr=3e-4))
    mcfg = Cfg(name="RigL", density=0.5, sparse_init="erdos-renyi-kernel", dense_gradients=True,
               growth_mode="absolute-gradient", prune_mode="magnitude", redistribution_mode="none",
               dense=False, prune_rate=0.1, decay_schedule="cosine", end_when=90, interval=10)
    mask = th.setup_mask(m, optim, mcfg)
    names = list(mask.mask_dict.keys())
    mask0 = np.packbits(np.concatenate([mask.mask_dict[n].numpy().ravel().astype(np.uint8) for n in names]))
    ls, rates, dens = [], [], []
    for i in range(120):
        ls.append(th.train_epoch(m, optim, grid_s, img_s, lr_scheduler=sched, mask=mask))
        if i <= mcfg.end_when and i % mcfg.interval == 0:
            mask.update_connections()
        rates.append(mask.prune_rate_decay.get_dr())
        dens.append(mask.stats.total_density)
    _, l, psnr, psnr8 = th.eval_epoch(m, grid_s, img_s)
    maskN = np.packbits(np.concatenate([mask.mask_dict[n].numpy().ravel().astype(np.uint8) for n in names]))
    np.savez_compressed(f"{OUT}/rigl_64x4_64.npz", init=p0, mask0=mask0, maskN=maskN,
                        losses=np.array(ls, np.float64), rates=np.array(rates), density=np.array(dens),
                        final=flat_params(m), psnr=psnr, psnr8=psnr8, mask_step=mask.mask_step,
                        img=img_s.numpy())
    print("rigl_64x4_64: psnr %.4f mask_step %d density %.4f" % (psnr, mask.mask_step, dens[-1]))


if __name__ == "__main__":
    main()
