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
    """Per-layer keep-probability eps * score, score = sum(shape) / prod(shape) (kernel form) or (n_in + n_out) / (n_in *
    n_out); eps is solved so that the expected number of ones equals the budget, and while the largest score times eps
    exceeds 1 the layers holding that score are made dense and eps is solved again (reference behaviour:
    init_scheme.py:66-129; the integer truncations of the budget terms are the reference's and decide eps to the last bit)."""
    sizes = {n: int(np.prod(sh)) for n, sh in shapes.items()}
    score = {n: float(np.sum(sh) / np.prod(sh)) if is_kernel else (sh[0] + sh[1]) / (sh[0] * sh[1]) for n, sh in shapes.items()}
    full = set()
    while True:
        budget, weight_sum = 0.0, 0.0
        for n in shapes:                                     # (accumulated in layer order: float sums)
            if n in full:
                budget -= int(sizes[n] * (1 - density))      # a dense layer spends the zeros it was entitled to
            else:
                budget += int(sizes[n] * density)
                weight_sum += score[n] * sizes[n]
        eps = budget / weight_sum
        open_scores = [score[n] for n in shapes if n not in full]
        top = max(open_scores)
        if top * eps <= 1:
            return {n: 1.0 if n in full else eps * score[n] for n in shapes}
        full.update(n for n in shapes if n not in full and score[n] == top)


def _draw_masks(masking, make_mask, skip_first: bool = False):
    """Walk the masked parameters in named_parameters() order, install make_mask(name, parameter) and book the counts."""
    for pos, (pname, par) in enumerate(masking.module.named_parameters()):
        if skip_first and pos == 0:
            masking.mask_dict.pop(pname, None)
            continue
        if pname in masking.mask_dict:
            mk = make_mask(pname, par)
            masking.mask_dict[pname] = mk
            masking.baseline_nonzero += int((mk != 0).sum().item())
            masking.total_params += par.numel()


def erk_init(masking, is_kernel: bool = True, **_):
    """init_scheme.py:147-158: Bernoulli(prob_layer) masks drawn on the CPU generator in named_parameters() order."""
    probs = erdos_renyi_densities({n: m.shape for n, m in masking.mask_dict.items()}, masking.density, is_kernel)
    _draw_masks(masking, lambda pname, par: (torch.rand(par.shape) < probs[pname]).float())
    masking.erk_probs = probs


def random_init(masking, **_):
    """init_scheme.py:181-206: every layer keeps `density` of its weights at random; the FIRST parameter is taken out of the
    mask set altogether (stays dense)."""
    _draw_masks(masking, lambda pname, par: (torch.rand(par.shape) < masking.density).float(), skip_first=True)


def resume_init(masking, **_):
    """init_scheme.py:209-228: mask = currently non-zero weights."""
    _draw_masks(masking, lambda pname, par: (par != 0.0).float().cpu())


init_registry = {"erdos-renyi-kernel": erk_init, "erdos-renyi": lambda m, **kw: erk_init(m, is_kernel=False, **kw),
                 "random": random_init, "resume": resume_init}


# ---------------------------------------------------------------------------------------------
# funcs/prune.py:24-51  magnitude pruning
# ---------------------------------------------------------------------------------------------
def magnitude_prune(masking, mask: torch.Tensor, weight: torch.Tensor, name: str) -> torch.Tensor:
    """prune.py:24-51: the ceil(rate * live) smallest |w| among the live weights go (the already-masked weights are zero
    and sort first, hence the offset by the dead count)."""
    drop = math.ceil(masking.name2prune_rate[name] * masking.stats.nonzeros_dict[name])
    if drop:
        order = torch.sort(weight.data.abs().view(-1)).indices
        mask.data.view(-1)[order[: masking.stats.zeros_dict[name] + drop]] = 0.0
    return mask


def global_magnitude_prune(masking) -> int:
    """One global |w| threshold over all masked layers (reference behaviour: prune.py:54-104): the threshold kept on the
    Masking object is nudged multiplicatively until the number of weights it removes is within `tolerance` of
    ceil(prune_rate * baseline_nonzero), or the count has not moved for 10 tries; the masks then become |w| > threshold.

    Own form: the magnitudes of every masked layer are sorted ONCE; "how many survive threshold t" is then a binary search
    (N - searchsorted(t, right=True)) instead of a pass over every layer per trial - same integer counts, hence the same
    threshold sequence and the same masks as the reference's per-layer sums (tests/golden/masking_pruning.npz)."""
    target = math.ceil(masking.prune_rate * masking.baseline_nonzero)
    if target <= 0:
        return 0
    layers = [(n, w) for n, w in masking.module.named_parameters() if n in masking.mask_dict]
    magnitudes = torch.sort(torch.cat([w.data.abs().reshape(-1) for _, w in layers])).values
    alive = sum(masking.stats.nonzeros_dict[n] for n, _ in layers)

    def removed_by(threshold: float) -> int:
        t = torch.tensor(threshold, dtype=magnitudes.dtype, device=magnitudes.device)      # (the comparison is in the weights' dtype)
        survivors = magnitudes.numel() - int(torch.searchsorted(magnitudes, t, right=True))
        return alive - survivors

    slack = target * masking.tolerance
    step, removed, previous, stalled = masking.increment, 0, 0, 0
    while abs(removed - target) > slack:
        removed = removed_by(masking.prune_threshold)
        stalled = stalled + 1 if removed == previous else 0
        if stalled == 10:
            break
        previous = removed
        if removed > target * (1.0 + masking.tolerance):        # too many gone: lower the bar
            masking.prune_threshold *= 1.0 - step
            step *= 0.99
        elif removed < target * (1.0 - masking.tolerance):      # too few: raise it
            masking.prune_threshold *= 1.0 + step
            step *= 0.99
    for name, weight in layers:
        masking.mask_dict[name][:] = weight.data.abs() > masking.prune_threshold
    return int(removed)


prune_registry = {"magnitude": magnitude_prune, "global-magnitude": global_magnitude_prune}


# ---------------------------------------------------------------------------------------------
# funcs/grow.py:58-97  absolute-gradient growth (grown weights start at 0)
# ---------------------------------------------------------------------------------------------
def abs_grad_growth(masking, name: str, total_regrowth: int, weight: torch.Tensor) -> torch.Tensor:
    live = masking.mask_dict[name].data.bool()
    if bool(live.all()):
        return live
    candidates = weight.grad.abs() * (~live).to(weight.grad.dtype)          # |dL/dw| where the weight is masked out
    pick = torch.sort(candidates.flatten(), descending=True).indices[: int(total_regrowth)]
    live.view(-1)[pick] = True
    weight.data.view(-1)[pick] = 0.0
    return live


def momentum_growth(masking, name: str, total_regrowth: int, weight: torch.Tensor) -> torch.Tensor:
    """grow.py:25-55: grow where |Adam momentum| = |m / (sqrt(v) + 1e-8)| is largest among masked-out entries
    (the weights themselves are NOT reset by this mode)."""
    live = masking.mask_dict[name].data.bool()
    mom = masking.get_momentum_for_weight(weight)
    candidates = (mom * (~live).to(mom.dtype)).abs()
    live.view(-1)[torch.sort(candidates.flatten(), descending=True).indices[: int(total_regrowth)]] = True
    return live


def random_growth(masking, name: str, total_regrowth: int, weight: torch.Tensor) -> torch.Tensor:
    """grow.py:100-136: Bernoulli(total_regrowth / #zeros) growth among masked-out entries (device RNG of the
    mask tensor); grown and still-masked weights are zeroed."""
    live = masking.mask_dict[name].data.bool()
    free = int((~live).sum().item())
    if free == 0:
        return live
    born = torch.zeros_like(live)
    born[~live] = torch.rand_like(born[~live].float()) < total_regrowth / free
    live = live | born
    weight.data[born] = 0.0
    weight.data[~live] = 0.0
    return live


def no_growth(masking, name: str, total_regrowth: int, weight: torch.Tensor) -> torch.Tensor:
    return masking.mask_dict[name].data.bool()


grow_registry = {"absolute-gradient": abs_grad_growth, "momentum": momentum_growth, "random": random_growth,
                 "none": no_growth}


# ---------------------------------------------------------------------------------------------
# funcs/redistribute.py:60-94  ('none' keeps the per-layer non-zero counts)
# ---------------------------------------------------------------------------------------------
def nonzero_statistic(masking, name, weight, mask) -> float:
    return (weight != 0.0).sum().item()


def momentum_statistic(masking, name, weight, mask) -> float:
    """redistribute.py:18-37: mean |Adam momentum| over the active weights."""
    return torch.abs(masking.get_momentum_for_weight(weight)[mask.bool()]).mean().item()


def grad_statistic(masking, name, weight, mask) -> float:
    return torch.abs(weight.grad[mask.bool()]).mean().item()


redistribute_registry = {"none": nonzero_statistic, "nonzero": nonzero_statistic, "momentum": momentum_statistic,
                         "grad": grad_statistic}


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


class LinearDecay:
    """decay.py:73-112."""
    mode = "current"

    def __init__(self, prune_rate: float = 0.3, T_max: int = 1000):
        self._step, self.T_max = 0, T_max
        self.decrement = prune_rate / float(T_max)
        self.current_prune_rate = self.initial_prune_rate = prune_rate

    def step(self, step: int = -1, *_):
        if step >= 0:
            if self._step < self.T_max:
                self.current_prune_rate = self.initial_prune_rate - self.decrement * (step + 1)
                self._step = step + 1
            else:
                self._step = self.T_max
            return
        if self._step < self.T_max:
            self.current_prune_rate -= self.decrement
            self._step += 1

    def get_dr(self) -> float:
        return self.current_prune_rate


class MagnitudePruneDecay:
    """decay.py:115-158 (Zhu & Gupta cubic cumulative-sparsity schedule; prune rate = finite difference
    against the CURRENT sparsity handed in by Masking.step)."""
    mode = "cumulative"

    def __init__(self, initial_sparsity: float = 0.0, final_sparsity: float = 0.3, T_max: int = 30000,
                 T_start: int = 350, interval: int = 100):
        self.initial_sparsity, self.final_sparsity = initial_sparsity, final_sparsity
        self.T_max, self.T_start, self.interval = T_max, T_start, interval
        self.current_prune_rate, self._step = 0.0, 0

    def cumulative_sparsity(self, step):
        if step < self.T_start:
            return self.initial_sparsity
        if step < self.T_max:
            mul = (1 - (step - self.T_start) / (self.T_max - self.T_start)) ** 3
            return self.final_sparsity + (self.initial_sparsity - self.final_sparsity) * mul
        return self.final_sparsity

    def step(self, step: int = -1, current_sparsity=-1):
        if step == -1:
            step = self._step
        if current_sparsity == -1:
            current_sparsity = self.cumulative_sparsity(step - self.interval)
        self.current_prune_rate = max(self.cumulative_sparsity(step) - current_sparsity, 0)
        self._step = step + 1

    def get_dr(self) -> float:
        return self.current_prune_rate


decay_registry = {"cosine": CosineDecay, "linear": LinearDecay, "magnitude-prune": MagnitudePruneDecay}
