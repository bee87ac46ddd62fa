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
         forward_backward(sync=False)   enqueue forward + backward of the local rows (no host sync)
         grad_view() -> flat fp32       the backend's OWN gradient buffer, already scaled by 1/(3*H*W) of the
                                        full image; all-reduced in place
         sse_view()  -> float64[1]      local sum of squared residuals on the backend's device; all-reduced in place
         adam_step(lr)
       (SirenEngine created with row_begin/row_end satisfies it).  `n_values` = 3*H*W of the full image.

    One step = two collectives on the backend's own buffers (P floats + one double), no staging copies and no host
    synchronisation: the stream stays busy across steps, and the loss is read back only when asked for."""

    def __init__(self, backend, n_values: int, group=None):
        import torch.distributed as dist
        self.dist, self.backend, self.n_values, self.group = dist, backend, n_values, group

    def _all_reduce(self, t: torch.Tensor):
        if self.dist.get_backend(self.group) == "gloo" and t.is_cuda:   # rehearsal path: gloo reduces on the host
            host = t.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def step(self, lr: float, want_loss: bool = True):
        """One optimiser step on the full image; returns the global MSE (same on every rank) or None."""
        self.backend.forward_backward(sync=False)
        sse = self.backend.sse_view()
        self._all_reduce(self.backend.grad_view())
        self._all_reduce(sse)
        loss = float(sse.item()) / self.n_values if want_loss else None   # read before the next pass overwrites it
        self.backend.adam_step(lr)
        return loss
