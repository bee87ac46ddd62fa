#!/usr/bin/env python3
"""Golden vectors for the other masking modes, minted by running the REAL reference.

    python tests/golden/make_golden_masking.py

For each mode a short masked training run on the CPU; right before one `update_connections()` the full
state (weights, gradients, Adam moments, masks, decay state) is captured, and the state right after it.
Modes: SNFS (momentum growth + momentum redistribution, conf/masking/SNFS.yaml), Pruning (global-magnitude,
magnitude-prune decay, no growth, random init, conf/masking/Pruning.yaml), SET (magnitude + random growth),
plus the LinearDecay / MagnitudePruneDecay sequences.
"""
import importlib.util
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


class Cfg(dict):
    __getattr__ = dict.get


def flat(ts):
    return np.concatenate([t.detach().numpy().ravel() for t in ts]).astype(np.float32)


def main():
    sys.path.insert(0, REF)
    _stub("omegaconf", DictConfig=dict, OmegaConf=object)
    _stub("torch_optimizer", Shampoo=object)
    from implicit_image.utils import train_helper as th
    from implicit_image.pipeline.masking.funcs import decay
    spec = importlib.util.spec_from_file_location("ref_siren", f"{REF}/implicit_image/models/siren.py")
    siren = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(siren)
    sys.path.insert(0, os.path.dirname(OUT))
    from golden.make_golden import synthetic_image  # noqa: the same formula image

    hw = 32
    img = synthetic_image(hw, hw, seed=5)
    gh = torch.linspace(0, 1, hw)
    grid = torch.stack(torch.meshgrid(gh, gh, indexing="ij"), dim=-1)
    modes = {
        "snfs": Cfg(name="SNFS", density=0.3, sparse_init="erdos-renyi-kernel", dense_gradients=True,
                    growth_mode="momentum", prune_mode="magnitude", redistribution_mode="momentum", dense=False,
                    prune_rate=0.1, decay_schedule="cosine", end_when=60, interval=5),
        "pruning": Cfg(name="Pruning", density=1.0, sparse_init="random", final_density=0.5, dense_gradients=True,
                       growth_mode="none", prune_mode="global-magnitude", redistribution_mode="none", dense=False,
                       decay_schedule="magnitude-prune", start_when=5, end_when=60, interval=5),
        "set": Cfg(name="SET", density=0.5, sparse_init="erdos-renyi-kernel", dense_gradients=False,
                   growth_mode="random", prune_mode="magnitude", redistribution_mode="none", dense=False,
                   prune_rate=0.2, decay_schedule="linear", end_when=60, interval=5),
    }
    for tag, mcfg in modes.items():
        torch.manual_seed(0)
        m = siren.Siren(depth=4, hidden_size=64, first_omega_0=50, hidden_omega_0=30)
        optim, sched = th.get_optimizer_lr_scheduler(m, Cfg(name="adam", lr=3e-4))
        mask = th.setup_mask(m, optim, mcfg)
        names = [n for n, _ in m.named_parameters() if n in mask.mask_dict]
        out = {"mask_names": np.array(names)}
        out["mask_init"] = np.packbits(np.concatenate([mask.mask_dict[n].numpy().ravel().astype(np.uint8) for n in names]))
        rates = []
        for i in range(13):
            th.train_epoch(m, optim, grid, img, lr_scheduler=sched, mask=mask)
            rates.append(mask.prune_rate)
            if i in (5, 10):
                # state before the update
                pre = f"u{i}_"
                out[pre + "w"] = flat(m.parameters())
                out[pre + "g"] = flat([p.grad for p in m.parameters()])
                out[pre + "m"] = flat([optim.state[p]["exp_avg"] for p in m.parameters()])
                out[pre + "v"] = flat([optim.state[p]["exp_avg_sq"] for p in m.parameters()])
                out[pre + "mask"] = np.packbits(np.concatenate(
                    [mask.mask_dict[n].numpy().ravel().astype(np.uint8) for n in names]))
                out[pre + "mask_step"] = mask.mask_step
                out[pre + "rate"] = mask.prune_rate
                out[pre + "adjusted_growth"] = float(mask.adjusted_growth)
                out[pre + "adjustments"] = np.array(mask.adjustments, dtype=np.float64)
                out[pre + "prune_threshold"] = float(mask.prune_threshold)
                out[pre + "total_nonzero"] = mask.stats.total_nonzero
                out[pre + "total_zero"] = mask.stats.total_zero
                torch.manual_seed(1000 + i)          # pins random growth
                mask.update_connections()
                post = f"a{i}_"
                out[post + "w"] = flat(m.parameters())
                out[post + "m"] = flat([optim.state[p]["exp_avg"] for p in m.parameters()])
                out[post + "mask"] = np.packbits(np.concatenate(
                    [mask.mask_dict[n].numpy().ravel().astype(np.uint8) for n in names]))
                out[post + "mask_step"] = mask.mask_step
                out[post + "nnz"] = np.array([int(mask.mask_dict[n].sum().item()) for n in names])
                out[post + "adjusted_growth"] = float(mask.adjusted_growth)
                out[post + "prune_threshold"] = float(mask.prune_threshold)
                out[post + "density"] = mask.stats.total_density
            elif i <= mcfg.end_when and i % mcfg.interval == 0:
                mask.update_connections()
        out["rates"] = np.array(rates, dtype=np.float64)
        np.savez_compressed(f"{OUT}/masking_{tag}.npz", **out)
        print(tag, "rates", [round(r, 5) for r in rates[:7]], "nnz", out["a10_nnz"].tolist(), "density", out["a10_density"])

    # decay sequences
    lin = decay.LinearDecay(prune_rate=0.2, T_max=20)
    seq_l = []
    for s in range(30):
        lin.step(s)
        seq_l.append(lin.get_dr())
    mp = decay.MagnitudePruneDecay(final_sparsity=0.5, T_max=60, T_start=5, interval=5)
    seq_m = []
    for s in range(70):
        mp.step(s, 0.01 * s)
        seq_m.append(mp.get_dr())
    np.savez_compressed(f"{OUT}/decay_linear_magprune.npz", linear=np.array(seq_l), magprune=np.array(seq_m))


if __name__ == "__main__":
    main()
