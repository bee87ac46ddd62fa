"""Deep-Compression style k-means quantisation of Linear weights (reference:
pipeline/quant/kmeans.py:110-181, kmeans_helper.py:59-115).

Index path (labels -> uint8 in the container) is bit-exact given identical weights.  The centroid
update uses `scatter_mean`, a third-party op (torch_scatter, absent and unpinned): it is restated here
from its documented semantics, so centroid VALUES are "parity unpinned" at that boundary.

Engine coupling: the reference hangs a forward-pre-hook on every nn.Linear; the engine-backed Siren
never calls those modules, so `KmeansQuant` registers ONE model-level callback that re-clusters all
non-skipped layers in module order right before every engine pass (Siren.engine()), and one
post-backward callback for the centroid nudge (kmeans.py:163-181).
"""
from typing import List, Sequence

import torch
from torch import nn


def scatter_mean(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """torch_scatter.scatter_mean(src, index, dim=0): per-index mean of the rows of `src`; output has
    index.max()+1 rows; empty bins are 0 (count clamped to 1)."""
    n = int(index.max().item()) + 1
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    out.index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype, device=src.device)
    cnt.index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype, device=src.device))
    return out / cnt.clamp(min=1).reshape(-1, *([1] * (src.dim() - 1)))


def _sq_dist(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[n,f] x [m,f] -> [n,m] squared euclidean distances (kmeans_helper.py:10-22)."""
    return ((a[:, None, :] - b[None, :, :]) ** 2.0).sum(dim=-1)


def kmeans_fit(x: torch.Tensor, cluster_centers: torch.Tensor, tolerance: float = 1e-4, iter_limit: int = 5):
    """<= 5 Lloyd iterations from the given centres; stops when (sum_k ||dc_k||)^2 < tol
    (kmeans_helper.py:59-98)."""
    x = x.float()
    labels = None
    for _ in range(iter_limit):
        labels = torch.argmin(_sq_dist(x, cluster_centers), dim=1)
        new_centers = scatter_mean(x, labels)
        shift = torch.sqrt(torch.sum((cluster_centers - new_centers) ** 2, dim=1)).sum()
        cluster_centers = new_centers
        if shift ** 2 < tolerance:
            break
    return labels, cluster_centers


def kmeans_predict(x: torch.Tensor, cluster_centers: torch.Tensor) -> torch.Tensor:
    return torch.argmin(_sq_dist(x, cluster_centers), dim=1)        # first index wins ties


def find_centroids_native(weight: torch.Tensor, n_clusters: int, eng):
    """The same on the engine's stream WITHOUT a host synchronisation (sf_kmeans_fit, csrc/siren_kmeans.hip): returns
    (centroids zero-padded to n_clusters entries, n_centroids as a device int32, labels, new_weight).  The initial guess is
    torch.linspace over masked min / max, so it is bit-identical to the torch path's."""
    w = weight.detach().float().contiguous()
    nz = w != 0
    inf = torch.tensor(float("inf"), device=w.device)
    mn = torch.where(nz, w, inf).min()
    mx = torch.where(nz, w, -inf).max()
    guess = torch.linspace(mn, mx, n_clusters - 1, device=w.device, dtype=w.dtype)
    return eng.kmeans_fit(w, guess)


def find_centroids(weight: torch.Tensor, n_clusters: int):
    """(centroids, labels, new_weight) of one weight tensor (kmeans.py:110-150): linspace(min,max)
    guess over the NON-ZERO weights with n_clusters-1 centres, 0 prepended, torch.unique, sorted by
    |c|, labels = nearest centroid of every weight (zeros included)."""
    shape = weight.shape
    w = weight.reshape(-1, 1)
    nz = w[w != 0].reshape(-1, 1)
    guess = torch.linspace(nz.min(), nz.max(), n_clusters - 1, device=w.device, dtype=w.dtype).reshape(-1, 1)
    _, centroids = kmeans_fit(nz, guess)
    centroids = torch.cat((torch.zeros_like(centroids)[:1], centroids))
    centroids = torch.unique(centroids)
    _, order = torch.sort(centroids.abs())
    centroids = centroids[order]
    labels = kmeans_predict(w, centroids.reshape(-1, 1)).reshape(*shape)
    return centroids, labels, centroids[labels]


class KmeansQuant:
    def __init__(self, model: nn.Module, optim, bits: int = 5, skip_ll: Sequence[str] = ("layers.0.linear", "layers.7.linear")):
        self.model, self.optim, self.bits, self.skip_ll = model, optim, bits, list(skip_ll)
        self._targets: List[nn.Linear] = [m for n, m in model.named_modules()
                                         if isinstance(m, nn.Linear) and n not in self.skip_ll]
        self._pre = self.kmeans_modify_weights
        self._post = self.scalar_quantization
        model.pre_pass_callbacks.append(self._pre)
        model.post_backward_callbacks.append(self._post)

    @property
    def n_clusters(self) -> int:
        return 2 ** self.bits

    @property
    def learning_rate(self) -> float:
        return self.optim.defaults["lr"]

    @torch.no_grad()
    def kmeans_modify_weights(self):
        """forward-pre-hook of the reference, for every quantised layer in module order (kmeans.py:66-72)."""
        import os
        eng = getattr(self.model, "_engine", None)
        if os.environ.get("SIREN_FIT_NATIVE_KMEANS", "1") == "0":       # A/B knob: the torch host mirror
            eng = None
        for m in self._targets:
            if eng is not None and m.weight.is_cuda and self.n_clusters <= 512:
                # native path: five small kernels per layer on the engine's stream, no host sync; the centroid tensor keeps
                # its full 2^bits length (zero padded) until update_weights() trims it to the count the device reports
                centroids, m.n_centroids, labels, new_weight = find_centroids_native(m.weight.data, self.n_clusters, eng)
            else:
                centroids, labels, new_weight = find_centroids(m.weight.data, self.n_clusters)
                m.n_centroids = None
            m.labeled_weight, m.centroids = labels, centroids
            m.weight.data.copy_(new_weight)

    @torch.no_grad()
    def scalar_quantization(self):
        """backward hook (kmeans.py:163-181): centroids -= lr * scatter_add(labels, dL/dW)."""
        for m in self._targets:
            dw = torch.zeros_like(m.centroids)
            dw.scatter_add_(0, m.labeled_weight.flatten(), m.weight.grad.flatten())
            m.centroids = m.centroids - self.learning_rate * dw

    def remove_hooks(self):
        for lst, cb in ((self.model.pre_pass_callbacks, self._pre), (self.model.post_backward_callbacks, self._post)):
            if cb in lst:
                lst.remove(cb)

    @torch.no_grad()
    def update_weights(self):
        """Freeze centroids + labels as (non-trainable) parameters and write the codebook weights
        (kmeans.py:73-100)."""
        self.remove_hooks()
        for m in self._targets:
            centroids, labels = m.centroids, m.labeled_weight
            if getattr(m, "n_centroids", None) is not None:
                centroids = centroids[:int(m.n_centroids.item())]     # (the one host read of the quantise phase: at its end)
                m.n_centroids = None
            m.centroids = nn.Parameter(centroids, requires_grad=False)
            m.labeled_weight = nn.Parameter(labels, requires_grad=False)
            m.weight.data.copy_(centroids[labels])
