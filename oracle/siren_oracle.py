"""CPU oracle for the SIREN fitting hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (implicit-image-compression_amd/) never does and fails loudly without its HIP library.

This is a plain fp32 restatement (torch CPU ops, explicit forward/backward, no autograd) of the
reference algorithm.  Each function cites the reference lines it follows (paths relative to the
reference tree).  Parity status: PINNED — checked against golden vectors minted by running the real
reference in the build container (tests/golden/make_golden.py -> tests/golden/*.npz; see
tests/test_oracle_golden.py).
"""
import math
from typing import List, Sequence, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------------------------
# data.py:78-88  get_grid
# --------------------------------------------------------------------------------------------
def grid_vectors(height: int, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """The two 1-D vectors get_grid() meshes (data.py:82-83).  torch.linspace is kept as the
    generator: its fp32 output is not reproduced by i/(n-1)."""
    return torch.linspace(0, 1, height), torch.linspace(0, 1, width)


def get_grid(height: int, width: int) -> torch.Tensor:
    """data.py:78-88: [H, W, 2], 'ij' meshgrid of the two linspace vectors, stacked last."""
    gh, gw = grid_vectors(height, width)
    return torch.stack((gh[:, None].expand(height, width), gw[None, :].expand(height, width)), dim=-1).contiguous()


# --------------------------------------------------------------------------------------------
# siren.py:35-54, 72-121  parameter shapes, init and draw order
# --------------------------------------------------------------------------------------------
def layer_dims(hidden: int, depth: int, in_features: int = 2, out_features: int = 3) -> List[Tuple[int, int]]:
    """(in, out) per Linear layer; depth counts first and last (siren.py:90-118)."""
    dims = [in_features] + [hidden] * (depth - 1) + [out_features]
    return [(dims[i], dims[i + 1]) for i in range(depth)]


def siren_init(hidden: int, depth: int, first_omega_0: float = 50.0, hidden_omega_0: float = 30.0,
               seed: int = None, generator: torch.Generator = None) -> List[torch.Tensor]:
    """Returns [W0, b0, W1, b1, ...] exactly as `torch.manual_seed(seed); Siren(...)` produces them.

    Draw order per layer (siren.py:37,44-51): nn.Linear's kaiming-uniform weight (discarded),
    nn.Linear's bias U(+-1/sqrt(in)) (kept), then weight.uniform_(+-bound) with
    bound = 1/in (first layer) or sqrt(6/in)/omega_0 (all others, incl. the last; the last layer is
    built with omega_0 = hidden_omega_0, siren.py:110-117)."""
    if seed is not None:
        torch.manual_seed(seed)
    params = []
    for i, (fin, fout) in enumerate(layer_dims(hidden, depth)):
        torch.empty(fout, fin).uniform_(-1.0, 1.0, generator=generator)  # kaiming draw, overwritten
        bb = 1.0 / math.sqrt(fin)
        b = torch.empty(fout).uniform_(-bb, bb, generator=generator)
        bound = 1.0 / fin if i == 0 else float(np.sqrt(6 / fin) / hidden_omega_0)
        w = torch.empty(fout, fin).uniform_(-bound, bound, generator=generator)
        params += [w, b]
    return params


def flatten(params: Sequence[torch.Tensor]) -> np.ndarray:
    return np.concatenate([p.detach().cpu().numpy().ravel() for p in params]).astype(np.float32)


def unflatten(flat, hidden: int, depth: int) -> List[torch.Tensor]:
    flat = torch.as_tensor(np.asarray(flat, dtype=np.float32))
    out, off = [], 0
    for fin, fout in layer_dims(hidden, depth):
        out.append(flat[off:off + fin * fout].reshape(fout, fin).clone()); off += fin * fout
        out.append(flat[off:off + fout].clone()); off += fout
    assert off == flat.numel()
    return out


# --------------------------------------------------------------------------------------------
# siren.py:56-68, 123-134  forward ; train_helper.py:147-161  mse + backward
# --------------------------------------------------------------------------------------------
def forward(params: Sequence[torch.Tensor], grid: torch.Tensor, first_omega_0: float = 50.0,
            hidden_omega_0: float = 30.0, keep: bool = False, outermost_linear: bool = True):
    """Siren.forward: flatten grid, x = (x-0.5)*2, Linear -> sin(omega*z) per layer (last layer linear when
    outermost_linear=True, the reference's configuration; sine with omega_0 = hidden_omega_0 otherwise,
    siren.py:110-117), out/2 + 0.5, reshape [H, W, 3]."""
    h, w, _ = grid.shape
    x = (grid.reshape(-1, 2) - 0.5) * 2
    depth = len(params) // 2
    acts, zs = [x], []
    for l in range(depth):
        z = torch.addmm(params[2 * l + 1], x, params[2 * l].t())
        om = first_omega_0 if l == 0 else hidden_omega_0
        x = torch.sin(z * om) if (l < depth - 1 or not outermost_linear) else z
        if keep:
            zs.append(z); acts.append(x)
    pred = (x / 2 + 0.5).reshape(h, w, -1)
    return (pred, acts, zs) if keep else pred


def loss_and_grads(params: Sequence[torch.Tensor], grid: torch.Tensor, img: torch.Tensor,
                   first_omega_0: float = 50.0, hidden_omega_0: float = 30.0, n_total: int = None,
                   outermost_linear: bool = True):
    """F.mse_loss(pred, img) (mean over 3*N, train_helper.py:151-154) and its gradient w.r.t. every
    parameter, by explicit back-propagation (what autograd computes at train_helper.py:159-161).
    n_total: pixel count of the FULL image when `grid`/`img` are a row shard (pixel-split mode)."""
    pred, acts, zs = forward(params, grid, first_omega_0, hidden_omega_0, keep=True, outermost_linear=outermost_linear)
    depth = len(params) // 2
    n = grid.shape[0] * grid.shape[1]
    n_total = n if n_total is None else n_total
    resid = pred.reshape(n, -1) - img.reshape(n, -1)
    sse = float((resid.double() ** 2).sum())
    delta = resid * (2.0 / (3.0 * n_total)) * 0.5      # d mse/d pred * d(out/2+0.5)/d out
    if not outermost_linear:
        delta = delta * (hidden_omega_0 * torch.cos(zs[depth - 1] * hidden_omega_0))
    grads = [None] * (2 * depth)
    for l in range(depth - 1, -1, -1):
        grads[2 * l] = delta.t() @ acts[l]
        grads[2 * l + 1] = delta.sum(0)
        if l > 0:
            om = first_omega_0 if l - 1 == 0 else hidden_omega_0
            delta = (delta @ params[2 * l]) * (om * torch.cos(zs[l - 1] * om))
    return sse / (3.0 * n_total), sse, grads


# --------------------------------------------------------------------------------------------
# torch.optim.Adam (selected at train_helper.py:72-78, conf/optim/adam.yaml) + StepLR (:80-84)
# --------------------------------------------------------------------------------------------
class Adam:
    """Single-tensor torch.optim.Adam arithmetic (torch 2.x op order), defaults beta=(0.9,0.999), eps=1e-8."""

    def __init__(self, params: Sequence[torch.Tensor], lr: float = 3e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    def step(self, params, grads, lr: float = None):
        lr = self.lr if lr is None else lr
        b1, b2 = self.betas
        self.t += 1
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        step_size, bc2s = lr / bc1, math.sqrt(bc2)
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            p.addcdiv_(m, (v.sqrt() / bc2s).add_(self.eps), value=-step_size)


def step_lr(base_lr: float, t: int, step_size: int = 2000, gamma: float = 0.5) -> float:
    """StepLR(optim, 2000, 0.5) stepped once per iteration (train_helper.py:80-84,183-184):
    learning rate used BY iteration t (0-based)."""
    return base_lr * gamma ** (t // step_size)


# --------------------------------------------------------------------------------------------
# train_helper.py:132-185 train_epoch ; :41-59 eval_epoch
# --------------------------------------------------------------------------------------------
def train_epoch(params, opt: Adam, grid, img, t: int, base_lr: float = 3e-4, masks=None,
                first_omega_0: float = 50.0, hidden_omega_0: float = 30.0, lr_step: int = 2000) -> float:
    """One full-batch optimiser step; returns the pre-step loss like train_loss.item().
    masks: optional list aligned with params (None for biases) applied after the step
    (Masking.step -> apply_mask, masking/core.py:671-688,271-279)."""
    loss, _, grads = loss_and_grads(params, grid, img, first_omega_0, hidden_omega_0)
    opt.step(params, grads, lr=step_lr(base_lr, t, lr_step))
    if masks is not None:
        for p, mk in zip(params, masks):
            if mk is not None:
                p.mul_(mk)
    train_epoch.last_grads = grads
    return loss


def eval_epoch(params, grid, img, first_omega_0: float = 50.0, hidden_omega_0: float = 30.0):
    """eval_epoch (train_helper.py:41-59): pred, mse, PSNR = 10 log10(1/mse), PSNR_8bit with
    truncating .int() casts of img*255 and pred*255."""
    pred = forward(params, grid, first_omega_0, hidden_omega_0)
    return (pred,) + metrics(pred, img)


def metrics(pred: torch.Tensor, img: torch.Tensor):
    mse = torch.nn.functional.mse_loss(pred, img)
    psnr = 10 * torch.log10(1 / mse)
    mse8 = (((img * 255).int() - (pred * 255).int()) ** 2).float().mean()
    psnr8 = 10 * torch.log10(255 ** 2 / mse8)
    return mse.item(), psnr.item(), psnr8.item()


def synthetic_image(height: int, width: int, seed: int = 1234, noise: float = 0.05) -> torch.Tensor:
    """SURVEY.md §8(d) formula image (the one the golden vectors were minted on): sinusoids plus
    seeded uniform noise of amplitude `noise` (0.05 in every round-1 fixture), clamped to [0,1]."""
    ys = torch.linspace(0, 1, height)[:, None].expand(height, width)
    xs = torch.linspace(0, 1, width)[None, :].expand(height, width)
    kx = torch.tensor([1.0, 2.0, 3.0])
    ky = torch.tensor([3.0, 1.0, 2.0])
    img = 0.5 + 0.25 * torch.sin(12 * xs[..., None] * kx) + 0.25 * torch.cos(9 * ys[..., None] * ky)
    g = torch.Generator().manual_seed(seed)
    img = img + noise * (torch.rand(height, width, 3, generator=g) * 2 - 1)
    return img.clamp(0, 1).float().contiguous()


def nonsmooth_image(height: int, width: int, seed: int = 77) -> torch.Tensor:
    """Procedural test image with the content a photograph has and the formula image lacks (round-3 parity fixtures):
    hard step edges (axis-aligned bars and a diagonal edge), a region saturated at exactly 0 and one at exactly 1
    (the clamp is active there), a one-pixel checkerboard (the highest spatial frequency the grid carries), smooth
    shading elsewhere, and ~0.1 % isolated outlier pixels flipped to the far end of the range (heavy-tailed residuals:
    the fp8 delta scale of a chunk is derived from its rms, the outliers sit 30-100x above it late in a fit).
    Deterministic in (height, width, seed); independent of the reference."""
    ys = torch.linspace(0, 1, height)[:, None].expand(height, width)
    xs = torch.linspace(0, 1, width)[None, :].expand(height, width)
    ch = torch.tensor([1.0, 0.8, 0.6])
    img = (0.45 + 0.2 * torch.sin(7.0 * xs + 3.0 * ys)[..., None] * ch + 0.15 * (ys - 0.5)[..., None]).clone()
    # vertical bars (step edges), upper left quadrant
    bars = ((xs * 16).floor() % 2 == 0) & (xs < 0.5) & (ys < 0.4)
    img[bars] = torch.tensor([0.85, 0.2, 0.3])
    # diagonal edge, lower left
    diag = (ys > 0.55) & (xs < 0.45) & (ys - 0.55 > 0.8 * xs)
    img[diag] = img[diag] * 0.25
    # saturated regions: smooth ramps pushed through the clamp
    sat_hi = (xs > 0.6) & (ys < 0.3)
    img[sat_hi] = (0.9 + 0.6 * (xs[sat_hi] - 0.6) / 0.4)[..., None].expand(-1, 3)
    sat_lo = (xs > 0.6) & (ys > 0.7)
    img[sat_lo] = (0.1 - 0.5 * (ys[sat_lo] - 0.7) / 0.3)[..., None].expand(-1, 3)
    # one-pixel checkerboard, centre right
    ii = torch.arange(height)[:, None].expand(height, width)
    jj = torch.arange(width)[None, :].expand(height, width)
    chk = (xs > 0.55) & (xs < 0.95) & (ys > 0.35) & (ys < 0.65)
    img[chk] = (0.25 + 0.5 * ((ii[chk] + jj[chk]) % 2).float())[..., None].expand(-1, 3)
    img = img.clamp(0, 1)
    # sparse outliers
    g = torch.Generator().manual_seed(seed)
    out = torch.rand(height, width, generator=g) < 1e-3
    img[out] = (img[out] < 0.5).float()
    return img.float().contiguous()
