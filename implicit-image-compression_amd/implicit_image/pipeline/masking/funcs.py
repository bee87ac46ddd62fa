"""Topology functions of the RigL path: ERK density split, magnitude pruning, |gradient| growth,
cosine prune-rate decay.  Index arithmetic follows the reference line by line in MEANING (citations
below, paths under implicit_image/pipeline/masking/) so that, given identical inputs, the masks
are bit-identical; the code is written against plain tensors so it runs on CPU (tests) and GPU."""
import math
from typing import Dict

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------
# funcs/init_scheme.py:40-158  Erdos-Renyi(-Kernel) densities and random masks
# ---------------------------------------------------------------------------------------------
def erdos_renyi_densities(shapes: Dict[str, torch.Size], density: float, is_kernel: bool = True) -> Dict[str, float]:
    """Per-layer keep-probability eps * (sum(shape) / prod(shape)); layers whose probability would
    exceed 1 are made dense one maximum at a time and eps is re-solved (init_scheme.py:66-129)."""
    dense = set()
    while True:
        divisor, rhs, raw = 0.0, 0.0, {}
        for name, shape in shapes.items():
            n_param = int(np.prod(shape))
            n_zeros = int(n_param * (1 - density))
            n_ones = int(n_param * density)
            if name in dense:
                rhs -= n_zeros
            else:
                rhs += n_ones
                if is_kernel:
                    raw[name] = float(np.sum(shape) / np.prod(shape))
                else:
                    n_in, n_out = shape[:2]
                    raw[name] = (n_in + n_out) / (n_in * n_out)
                divisor += raw[name] * n_param
        eps = rhs / divisor
        max_prob = max(raw.values())
        if max_prob * eps > 1:
            dense.update(n for n, p in raw.items() if p == max_prob)
        else:
            break
    return {name: (1.0 if name in dense else eps * raw[name]) for name in shapes}


def erk_init(masking, is_kernel: bool = True, **_):
    """init_scheme.py:147-158: masks drawn on the CPU generator in named_parameters() order."""
    shapes = {n: masking.mask_dict[n].shape for n in masking.mask_dict}
    probs = erdos_renyi_densities(shapes, masking.density, is_kernel)
    for name, weight in masking.module.named_parameters():
        if name not in masking.mask_dict:
            continue
        masking.mask_dict[name] = (torch.rand(weight.shape) < probs[name]).float()
        masking.baseline_nonzero += int((masking.mask_dict[name] != 0).sum().item())
        masking.total_params += weight.numel()
    masking.erk_probs = probs


init_registry = {
    "erdos-renyi-kernel": erk_init,
    "erdos-renyi": lambda m, **kw: erk_init(m, is_kernel=False, **kw),
}


# ---------------------------------------------------------------------------------------------
# funcs/prune.py:24-51  magnitude pruning
# ---------------------------------------------------------------------------------------------
def magnitude_prune(masking, mask: torch.Tensor, weight: torch.Tensor, name: str) -> torch.Tensor:
    num_remove = math.ceil(masking.name2prune_rate[name] * masking.stats.nonzeros_dict[name])
    if num_remove == 0.0:
        return mask
    k = masking.stats.zeros_dict[name] + num_remove
    _, idx = torch.sort(torch.abs(weight.data.view(-1)))
    mask.data.view(-1)[idx[:k]] = 0.0
    return mask


prune_registry = {"magnitude": magnitude_prune}


# ---------------------------------------------------------------------------------------------
# funcs/grow.py:58-97  absolute-gradient growth (grown weights start at 0)
# ---------------------------------------------------------------------------------------------
def abs_grad_growth(masking, name: str, total_regrowth: int, weight: torch.Tensor) -> torch.Tensor:
    new_mask = masking.mask_dict[name].data.bool()
    if int((new_mask == 0).sum().item()) == 0:
        return new_mask
    grad = weight.grad * (new_mask == 0).to(weight.grad.dtype)
    _, idx = torch.sort(torch.abs(grad).flatten(), descending=True)
    sel = idx[: int(total_regrowth)]
    new_mask.data.view(-1)[sel] = True
    weight.data.view(-1)[sel] = 0.0
    return new_mask


grow_registry = {"absolute-gradient": abs_grad_growth}


# ---------------------------------------------------------------------------------------------
# funcs/redistribute.py:60-94  ('none' keeps the per-layer non-zero counts)
# ---------------------------------------------------------------------------------------------
def nonzero_statistic(masking, name, weight, mask) -> float:
    return (weight != 0.0).sum().item()


redistribute_registry = {"none": nonzero_statistic, "nonzero": nonzero_statistic}


# ---------------------------------------------------------------------------------------------
# funcs/decay.py:25-70  CosineDecay == CosineAnnealingLR closed form evaluated at `step`
# ---------------------------------------------------------------------------------------------
class CosineDecay:
    """prune_rate(t) = eta_min + (p0 - eta_min) * (1 + cos(pi * t / T_max)) / 2, frozen once the
    internal counter reaches T_max (decay.py:57-66; torch's CosineAnnealingLR._get_closed_form_lr)."""
    mode = "current"

    def __init__(self, prune_rate: float = 0.3, T_max: int = 1000, eta_min: float = 0.0, last_epoch: int = -1):
        self.base, self.T_max, self.eta_min = prune_rate, T_max, eta_min
        self._step = 0
        self._rate = prune_rate

    def _closed_form(self, t: int) -> float:
        return self.eta_min + (self.base - self.eta_min) * (1 + math.cos(math.pi * t / self.T_max)) / 2

    def step(self, step: int = -1, *_):
        if step >= 0:
            if self._step < self.T_max:
                self._rate = self._closed_form(step)
                self._step = step + 1
            else:
                self._step = self.T_max
            return
        if self._step < self.T_max:
            self._step += 1
            self._rate = self._closed_form(self._step)

    def get_dr(self) -> float:
        return self._rate


decay_registry = {"cosine": CosineDecay}
