"""Multi-GPU modes of the fit.  One process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

  * per-image sharding  — `shard_jobs`: independent fits round-robin over ranks, NO collective;
  * pixel-split         — `PixelSplitFit`: each rank owns a block of image rows of ONE large image;
                          loss and gradient are sums over pixels, so one all-reduce(sum) of the flat fp32
                          gradient (+ the SSE scalar) per step makes every replica take the identical
                          Adam step.  The reference has no counterpart (it is single-device).
"""
from typing import List, Sequence, Tuple

import torch


def shard_rows(height: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced row block [r0, r1) of rank `rank` (empty if height < world_size is avoided
    by giving the first height % world ranks one extra row)."""
    base, rem = divmod(height, world_size)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def shard_jobs(jobs: Sequence, world_size: int, rank: int) -> List:
    """Per-image sharding: job i runs on rank i % world_size."""
    return [j for i, j in enumerate(jobs) if i % world_size == rank]


class PixelSplitFit:
    """Data-parallel over pixels.  `backend` is anything with
         forward_backward() -> float          local sum of squared residuals; leaves the local gradient
         get_grads() -> flat fp32 tensor      local gradient, ALREADY scaled by 1/(3*H*W) of the full image
         set_grads(flat), adam_step(lr)
       (SirenEngine created with row_begin/row_end satisfies it).  `n_values` = 3*H*W of the full image."""

    def __init__(self, backend, n_values: int, group=None):
        import torch.distributed as dist
        self.dist, self.backend, self.n_values, self.group = dist, backend, n_values, group
        self._buf = None

    def step(self, lr: float) -> float:
        """One optimiser step on the full image; returns the global MSE (same on every rank)."""
        sse = self.backend.forward_backward()
        g = self.backend.get_grads()
        if self._buf is None:
            self._buf = torch.empty(g.numel() + 1, dtype=torch.float64 if g.device.type == "cpu" else torch.float32,
                                    device=g.device)
        self._buf[:-1].copy_(g)
        self._buf[-1] = sse
        if self.dist.get_backend(self.group) == "gloo" and self._buf.is_cuda:   # rehearsal path: gloo reduces on the host
            host = self._buf.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            self._buf.copy_(host)
        else:
            self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM, group=self.group)
        self.backend.set_grads(self._buf[:-1].to(g.dtype).contiguous())
        self.backend.adam_step(lr)
        return float(self._buf[-1].item()) / self.n_values
