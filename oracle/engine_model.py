"""Numerics model of the HIP engine — TEST INFRASTRUCTURE (same import rules as siren_oracle.py).

The engine differs from the fp32 reference ONLY by operand roundings; this module restates the
engine's arithmetic in torch CPU ops with every rounding point made explicit, so GPU tests can
separate "rounding" (engine == this model to ~1e-5) from "algorithm" (this model ~= fp32 oracle):

  forward  hidden / last GEMM operands rounded to fp16 (weights pre-scaled by 2^8) or bf16, f32 accumulate;
           layer 0 in f32; sine computed on z*omega/(2 pi) revolutions;
  spill    only the PHASE frac(z*omega/2pi) of the HIDDEN layers as unorm16 (x*65535, round-nearest-even);
           layer-0 phases are re-derived from the pixel coordinates in the backward (fp32, not quantised);
  backward all GEMM operands in the same 16-bit type T as the forward (fp16 by default, deltas pre-scaled
           by a power of two so they sit in fp16's normal range - modelled here as *2^20):
           delta_l = T((delta W*omega) * cos(2 pi q/65536)), activations re-derived as T(sin(2 pi q/65536)),
           backward weight image = T(W * omega_{l-1}); dL/dout = T(resid/(3N)); layer-0 coordinates split x = T(x) + T(x - T(x)).
"""
import math
from typing import Sequence

import torch

TWO_PI = 2.0 * math.pi


def _rt(x, kind):
    if kind == "f16":
        return x.to(torch.float16).float()
    if kind == "bf16":
        return x.to(torch.bfloat16).float()
    raise ValueError(kind)


def _phase_q(t):
    """unorm16 of frac(t): v_cvt_pknorm_u16_f32 = round-to-nearest-even of x*65535."""
    fr = t - torch.floor(t)
    return torch.round(fr * 65535.0).clamp_(0, 65535)


def loss_and_grads(params: Sequence[torch.Tensor], grid: torch.Tensor, img: torch.Tensor, fwd: str = "f16",
                   first_omega_0: float = 50.0, hidden_omega_0: float = 30.0):
    depth = len(params) // 2
    h, w, _ = grid.shape
    n = h * w
    ws = 256.0 if fwd == "f16" else 1.0
    x = (grid.reshape(-1, 2) - 0.5) * 2
    W0, b0 = params[0], params[1]
    z = torch.addcmul(torch.addcmul(b0, x[:, 0:1], W0[:, 0]), x[:, 1:2], W0[:, 1])   # fma order of k_fwd
    sc_first = torch.tensor(first_omega_0 / TWO_PI, dtype=torch.float32)
    hs = torch.tensor(hidden_omega_0 / TWO_PI, dtype=torch.float32)   # folded into the hidden forward images / biases
    t = z * sc_first
    ph0 = t - torch.floor(t)          # layer-0 phases are not spilled: k_bwd re-derives them from the coordinates
    q = [None]
    a = torch.sin(TWO_PI * t.double()).float()
    for l in range(1, depth - 1):
        t = _rt(a, fwd) @ _rt(params[2 * l] * hs, fwd).t() + params[2 * l + 1] * hs   # accumulator = phase (revolutions)
        q.append(_phase_q(t))
        a = torch.sin(TWO_PI * t.double()).float()
    L = depth - 1
    out = (_rt(a, fwd) @ _rt(params[2 * L] * ws, fwd).t() + params[2 * L + 1] * ws) * (1.0 / ws)
    pred = out * 0.5 + 0.5
    resid = pred - img.reshape(n, -1)
    sse = float((resid.double() ** 2).sum())
    gscale = torch.tensor(1.0 / (3.0 * n), dtype=torch.float32)
    bwd = fwd                      # the backward GEMM operands use the same 16-bit type as the forward
    delta = _rt(resid * gscale, bwd) if bwd == "bf16" else _rt(resid * gscale * 2.0 ** 20, bwd) * 2.0 ** -20
    grads = [None] * (2 * depth)
    for l in range(L, 0, -1):
        # backward phase decode: u dropped into a float mantissa -> u/65536 revolutions (k_bwd phase_rev_*)
        ph = ph0 if l - 1 == 0 else q[l - 1] * (1.0 / 65536.0)
        act = _rt(torch.sin(TWO_PI * ph.double()).float(), bwd)
        grads[2 * l] = delta.t() @ act
        grads[2 * l + 1] = delta.sum(0)
        om = first_omega_0 if l - 1 == 0 else hidden_omega_0
        G = delta @ _rt(params[2 * l] * om, bwd)             # omega of layer l-1 is folded into the 16-bit image
        delta = _rt(G * torch.cos(TWO_PI * ph.double()).float() * 2.0 ** 20, bwd) * 2.0 ** -20
    xh = _rt(x, bwd)
    xl = _rt(x - xh, bwd)
    grads[0] = delta.t() @ xh + delta.t() @ xl
    grads[1] = delta.sum(0)
    return sse / (3.0 * n), sse, grads, pred.reshape(h, w, -1)
