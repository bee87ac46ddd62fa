"""Masking: wraps a model's weights with 0/1 masks (reference: pipeline/masking/core.py).

Differences from the reference are deliberate and documented:
  * the per-step `apply_mask` (core.py:271-279) is executed INSIDE the engine's Adam kernel when the
    model is engine-bound; the masks are pushed to the engine whenever the topology changes;
  * the FLOP counter (pipeline/masking/counting/) is logging only and is not rebuilt, but the two
    `torch.rand(*input_size)` draws it makes on the CPU generator (core.py:371,384) ARE reproduced,
    because the ERK masks drawn between them depend on the generator position.
Supported registry keys: sparse_init erdos-renyi(-kernel) | random | resume; prune_mode magnitude |
global-magnitude; growth_mode absolute-gradient | momentum | random | none; redistribution_mode none |
nonzero | momentum | grad; decay cosine | linear | magnitude-prune.  (The struct-* modes act on 4-D conv
kernels and have no meaning for a SIREN; lottery-ticket init loads a pickle and is not provided.)
"""
import logging
from dataclasses import dataclass, field
from typing import Dict

import numpy as np
import torch
from torch import nn

from .funcs import grow_registry, init_registry, prune_registry, redistribute_registry


@dataclass
class LayerStats:
    variance_dict: Dict[str, float] = field(default_factory=dict)
    zeros_dict: Dict[str, int] = field(default_factory=dict)
    nonzeros_dict: Dict[str, int] = field(default_factory=dict)
    removed_dict: Dict[str, int] = field(default_factory=dict)
    total_variance: float = 0
    total_zero: int = 0
    total_nonzero: int = 0
    total_removed: int = 0

    @property
    def total_density(self) -> float:
        tot = self.total_zero + self.total_nonzero
        return self.total_nonzero / tot if tot else 0.0


class Masking:
    def __init__(self, optimizer, prune_rate_decay, density: float = 0.2, sparse_init: str = "random",
                 dense_gradients: bool = False, prune_mode: str = "magnitude", growth_mode: str = "momentum",
                 redistribution_mode: str = "momentum", input_size=(1, 3, 32, 32)):
        assert sparse_init in init_registry, f"Sparse init {sparse_init} not found. Available {init_registry.keys()}"
        assert growth_mode in grow_registry or growth_mode == "none", \
            f"Available growth modes: {','.join(grow_registry.keys())}"
        assert prune_mode in prune_registry, f"Available prune modes: {','.join(prune_registry.keys())}"
        assert redistribution_mode in redistribute_registry, \
            f"Available redistribute modes: {','.join(redistribute_registry.keys())}"
        self.optimizer, self.prune_rate_decay = optimizer, prune_rate_decay
        self.density, self.sparse_init, self.dense_gradients = density, sparse_init, dense_gradients
        self.prune_mode, self.growth_mode, self.redistribution_mode = prune_mode, growth_mode, redistribution_mode
        self.input_size = input_size
        self.mask_dict: Dict[str, torch.Tensor] = {}
        self.module = None
        self.mask_step = 0
        self.baseline_nonzero = 0
        self.total_params = 0
        self.adjusted_growth = 0
        self.adjustments = []
        self.name2prune_rate = {}
        self.stats = LayerStats()
        self._pushed_engine = None
        # global growth/prune state (core.py:151-156)
        self.prune_threshold, self.growth_threshold = 0.001, 0.001
        self.growth_increment, self.increment, self.tolerance = 0.2, 0.2, 1e-6

    # ---- properties -------------------------------------------------------------------------
    @property
    def prune_rate(self) -> float:
        return self.prune_rate_decay.get_dr()

    @property
    def global_prune(self) -> bool:
        return "global" in self.prune_mode

    def get_momentum_for_weight(self, weight) -> torch.Tensor:
        """Adam: m / (sqrt(v) + 1e-8); SGD: momentum buffer (core.py:474-493)."""
        st = self.optimizer.state[weight]
        if "exp_avg" in st:
            return st["exp_avg"] / (torch.sqrt(st["exp_avg_sq"]) + 1e-08)
        return st["momentum_buffer"]

    def calc_redistributed_densities(self):
        """Regrowth budget per layer (reference behaviour: core.py:299-360): proportional to the redistribution statistic,
        no layer above 99 % of its free slots, what a capped layer cannot take is shared out equally and the shares are
        re-capped until nothing is left over (at most 1000 rounds).

        Own form: a water-filling over arrays - every round adds the same share to all budgets, clips them at the caps and
        collects the clipped excess (summed in layer order, as the reference accumulates it, so the float sequence and
        therefore the rounded budgets equal tests/golden/masking_snfs.npz bit for bit)."""
        names = list(self.stats.variance_dict)
        if not names:
            return {}
        pool = self.stats.total_removed + self.adjusted_growth
        caps = [0.99 * (self.stats.zeros_dict[n] + self.stats.removed_dict[n]) for n in names]
        budget = [round(self.stats.variance_dict[n] * pool) for n in names]      # (ints until a share is added)
        share = 0
        for _ in range(1000):
            excess = 0
            for j, cap in enumerate(caps):
                want = budget[j] + share
                if want > cap:
                    excess += want - cap
                    want = cap
                budget[j] = want
            share = excess / len(names)
            if not excess > 0:
                break
        return dict(zip(names, budget))

    # ---- setup (core.py:220-248, 386-423) ---------------------------------------------------
    def add_module(self, module: nn.Module):
        self.module = module
        torch.rand(*self.input_size)            # RNG draw of the dense-FLOPs probe (core.py:229,371)
        for name, weight in module.named_parameters():
            self.mask_dict[name] = torch.zeros_like(weight, dtype=torch.float32, requires_grad=False)
        for name in [n for n in self.mask_dict if "bias" in n]:       # core.py:239-240
            self.mask_dict.pop(name)
        for name, sub in module.named_modules():                      # core.py:241-244
            if isinstance(sub, (nn.BatchNorm1d, nn.BatchNorm2d)):
                self.mask_dict.pop(name + ".weight", None)
        init_registry[self.sparse_init](self)
        self._to_module_device()
        self.apply_mask()
        self.stats.total_nonzero = self.baseline_nonzero
        self.stats.total_zero = self.total_params - self.baseline_nonzero
        torch.rand(*self.input_size)            # RNG draw of the sparse-FLOPs log line (core.py:248,384)
        logging.info(f"Achieved sparsity at init (w/o BN, bias): {self.baseline_nonzero / self.total_params:.4f}")

    def _masked(self):
        """(name, parameter, mask) of every masked parameter, in named_parameters() order - the order every per-layer loop
        of the reference walks (float accumulations below depend on it)."""
        for pname, par in self.module.named_parameters():
            mk = self.mask_dict.get(pname)
            if mk is not None:
                yield pname, par, mk

    def _to_module_device(self):
        for pname, par, mk in list(self._masked()):
            self.mask_dict[pname] = mk.to(par.device)

    def flat_mask(self) -> torch.Tensor:
        """Current masks as one flat 0/1 vector in named_parameters() order (ones for unmasked parameters)."""
        parts = []
        for name, weight in self.module.named_parameters():
            mk = self.mask_dict.get(name)
            parts.append((mk if mk is not None else torch.ones_like(weight)).reshape(-1).float())
        return torch.cat(parts)

    def _push_masks(self):
        """Hand the current masks to the engine."""
        eng = getattr(self.module, "_engine", None)
        if eng is None:
            return
        self.module.set_engine_masks(self.flat_mask())
        self._pushed_engine = eng

    # ---- per-step (core.py:271-289, 671-702) ------------------------------------------------
    @torch.no_grad()
    def apply_mask(self):
        for _, par, mk in self._masked():
            par.data.mul_(mk.to(par.dtype))
        self._push_masks()

    @torch.no_grad()
    def apply_mask_gradients(self):
        for _, par, mk in self._masked():
            if par.grad is not None:
                par.grad.mul_(mk)

    @torch.no_grad()
    def reset_momentum(self):
        for _, par, mk in self._masked():
            st = self.optimizer.state[par]
            for key in ("exp_avg", "exp_avg_sq"):
                if key in st:
                    st[key].mul_(mk)

    def step(self, scaler=None):
        """Optimiser step, then masks, then prune-rate decay (core.py:671-702).  `scaler` is accepted
        for signature parity; GradScaler is an exact no-op on the fp32 path."""
        eng = getattr(self.module, "_engine", None)
        if eng is not None and self._pushed_engine is not eng:
            self._push_masks()                  # engine was created after add_module()
        self.optimizer.step()
        if eng is None or not getattr(self.optimizer, "applies_engine_mask", False):
            self.apply_mask()                   # otherwise fused into the engine's Adam kernel
        if not self.dense_gradients:
            self.reset_momentum()
        if self.prune_rate_decay.mode == "cumulative":
            self.prune_rate_decay.step(self.mask_step, 1 - self.stats.total_density)
        else:
            self.prune_rate_decay.step(self.mask_step)
        self.mask_step += 1

    # ---- topology update (core.py:425-464, 250-269, 713-801) ----------------------------------
    def gather_statistics(self):
        """Per-layer redistribution statistic (normalised to sum 1), live and pruned counts (core.py:425-464)."""
        stat_of = redistribute_registry[self.redistribution_mode]
        share, live, dead = {}, {}, {}
        for pname, par, mk in self._masked():
            share[pname] = stat_of(self, pname, par, mk)
            live[pname] = int((mk == 1).sum().int().item())
            dead[pname] = int((mk == 0).sum().int().item())
        norm = 0.0
        for v in share.values():               # (summed in layer order; NaN statistics are skipped, as the reference does)
            if not np.isnan(v):
                norm += v
        assert norm, "Total variance is zero!"
        share = {k: v / norm for k, v in share.items()}
        self.stats = LayerStats(variance_dict=share, nonzeros_dict=live, zeros_dict=dead, total_variance=norm,
                                total_nonzero=sum(live.values()), total_zero=sum(dead.values()))

    def adjust_prune_rate(self):
        """Layers that are still less than 20 % sparse and hold more than an equal share of the statistic are not pruned
        beyond their own sparsity (core.py:250-269)."""
        rate, fair = self.prune_rate, 1.0 / max(len(self.stats.variance_dict), 1)
        for pname, mk in self.mask_dict.items():
            frac_dead = self.stats.zeros_dict[pname] / mk.numel()
            capped = frac_dead < 0.2 and fair / self.stats.variance_dict[pname] < 1.0
            self.name2prune_rate[pname] = min(frac_dead, rate) if capped else rate

    @torch.no_grad()
    def truncate_weights(self):
        self.gather_statistics()
        self.adjust_prune_rate()
        # 1. prune (core.py:713-745)
        prune_fn = prune_registry[self.prune_mode]
        if self.global_prune:
            self.stats.total_removed = prune_fn(self)
        else:
            for pname, par, mk in list(self._masked()):
                kept = prune_fn(self, mk, par, pname)
                gone = self.stats.nonzeros_dict[pname] - int(kept.sum().item())
                self.stats.removed_dict[pname] = gone
                self.stats.total_removed += gone
                self.mask_dict[pname] = kept
        # 2. grow (core.py:747-779): each layer regrows what it lost, or its share of the pool when a redistribution
        #    statistic is active
        if self.growth_mode == "none":
            total_nonzero_new = self.stats.total_nonzero - self.stats.total_removed
        else:
            grow_fn = grow_registry[self.growth_mode]
            budgets = (self.calc_redistributed_densities() if self.redistribution_mode not in ("nonzero", "none")
                       else self.stats.removed_dict)
            total_nonzero_new = 0
            for pname, par, _ in list(self._masked()):
                grown = grow_fn(self, pname, budgets[pname], par)
                total_nonzero_new += grown.sum().item()
                del self.mask_dict[pname]                 # (re-inserted at the end: the reference's dict order after an update)
                self.mask_dict[pname] = grown.float()
        self.apply_mask()
        if not self.dense_gradients:
            self.reset_momentum()
            self.apply_mask_gradients()
        self.mask_step += 1
        self.adjustments.append(self.baseline_nonzero - total_nonzero_new)
        self.adjusted_growth = 0.25 * self.adjusted_growth + 0.75 * self.adjustments[-1] + np.mean(self.adjustments)
        self.gather_statistics()

    def update_connections(self):
        self.truncate_weights()

    # ---- checkpoint surface (core.py:495-506, 660-669) -----------------------------------------
    def state_dict(self):
        return {"baseline_nonzero": self.baseline_nonzero, "masks": self.mask_dict, "mask_step": self.mask_step,
                "total_params": self.total_params}

    def load_state_dict(self, sd):
        self.baseline_nonzero, self.mask_step, self.total_params = sd["baseline_nonzero"], sd["mask_step"], sd["total_params"]
        self.mask_dict = dict(sd["masks"])
        self._to_module_device()
        self.apply_mask()
