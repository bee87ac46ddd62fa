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

  scratch=12 (the default for fp16 at hidden <= 256): as above, but the phases are spilled as BYTES round(t*256) mod 256
           and decoded as u/256 revolutions; the deltas stay 16-bit floats.

  scratch=8 (sf_config.scratch_format = 8; csrc/siren_s8.hip):
           phases spilled as bytes round(t*256) mod 256 and decoded as u/256 revolutions; dL/dout stored as
           fp16(resid * 2^10); per pixel chunk G = 2^floor(log2(FP8_TARGET / rms(resid))), FP8_TARGET = 0.5 (csrc kFp8Target); every hidden delta is
           e4m3(saturate(.)) in units of G (the first one picks up G / 2^10 together with its cosine);
           gradients of the layers below the last are scaled by float(1 / (G * 3N)), the last layer's by
           float(1 / (2^10 * 3N)).
"""
import math
from typing import Sequence

import torch

FP8_TARGET = 0.5      # csrc/siren_kernels.hip kFp8Target

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


def _e4m3(x):
    """OCP fp8 e4m3, round-nearest-even, saturating at +-448 (k_bwd8: v_med3_f32 + v_cvt_pk_fp8_f32)."""
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


PHASE_EPS = 2.0 ** -16     # phase_rev8: a decoded byte phase never sits on a zero of sin / cos (siren_kernels.hip kPhaseEps)


def _phase_q8(t):
    """phase byte of k_fwd<.., S8>: low mantissa byte of t + 1.5*2^15 = round-half-even(t * 256) mod 256."""
    return torch.remainder(torch.round(t * 256.0), 256.0)


def loss_and_grads(params: Sequence[torch.Tensor], grid: torch.Tensor, img: torch.Tensor, fwd: str = "f16",
                   first_omega_0: float = 50.0, hidden_omega_0: float = 30.0, scratch: int = 16,
                   n_total: int = None):
    if scratch == 8:
        return _loss_and_grads_s8(params, grid, img, first_omega_0, hidden_omega_0, n_total)
    if scratch not in (12, 16):
        raise ValueError(scratch)
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
        q.append(_phase_q8(t) * 256.0 if scratch == 12 else _phase_q(t))   # both decoded below as q / 65536 revolutions
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
        ph = ph0 if l - 1 == 0 else q[l - 1] * (1.0 / 65536.0) + (PHASE_EPS if scratch == 12 else 0.0)
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


def fp8_layer_scales(params, first_omega_0, hidden_omega_0):
    """csrc k_fp8_norms + k_fp8_links: link[l] = 2^-round(log2(gain_l)), gain_l = omega_{l-1} sqrt(0.5 ||W_l||_F^2 / n_in) (sums in double);
    inv[l] = 1 / prod_{m > l} link[m] multiplies the weight gradient of layer l."""
    depth = len(params) // 2
    link, inv = [1.0] * depth, [1.0] * depth
    S = 1.0
    for l in range(depth - 1, -1, -1):
        inv[l] = 1.0 / S
        if l == 0:
            break
        Wl = params[2 * l].double()
        om = first_omega_0 if l - 1 == 0 else hidden_omega_0
        gain = min(max(om * math.sqrt(0.5 * float((Wl * Wl).sum()) / Wl.shape[1]), 1e-6), 1e6)
        m, e = math.frexp(gain)                      # round(log2(gain)) as the kernel takes it: e - (m < sqrt(1/2))
        link[l] = 2.0 ** -(e - (1 if m < 0.70710678118654752440 else 0))
        S *= link[l]
    return link, inv


def _loss_and_grads_s8(params, grid, img, first_omega_0, hidden_omega_0, n_total=None):
    """The 8-bit scratch path (fp16 operands): see the module docstring.  One pixel chunk (what every test uses)."""
    depth = len(params) // 2
    h, w, _ = grid.shape
    n = h * w
    nv = float(img.shape[-1]) * float(n_total if n_total else n)      # out_features * H * W of the full image
    x = (grid.reshape(-1, 2) - 0.5) * 2
    W0, b0 = params[0], params[1]
    z = torch.addcmul(torch.addcmul(b0, x[:, 0:1], W0[:, 0]), x[:, 1:2], W0[:, 1])
    t = z * torch.tensor(first_omega_0 / TWO_PI, dtype=torch.float32)
    t0 = t
    ph0 = t - torch.floor(t)
    hs = torch.tensor(hidden_omega_0 / TWO_PI, dtype=torch.float32)
    q = [None]
    a = torch.sin(TWO_PI * t.double()).float()
    for l in range(1, depth - 1):
        t = _rt(a, "f16") @ _rt(params[2 * l] * hs, "f16").t() + params[2 * l + 1] * hs
        q.append(_phase_q8(t))
        a = torch.sin(TWO_PI * t.double()).float()
    L = depth - 1
    out = (_rt(a, "f16") @ _rt(params[2 * L] * 256.0, "f16").t() + params[2 * L + 1] * 256.0) * (1.0 / 256.0)
    pred = out * 0.5 + 0.5
    resid = pred - img.reshape(n, -1)
    sse = float((resid.double() ** 2).sum())
    S0 = 1024.0
    dlast = _rt(resid * S0, "f16")
    rms = min(max(math.sqrt(sse / (img.shape[-1] * n)), 1e-12), 4.0)
    G = 2.0 ** math.floor(math.log2(FP8_TARGET / rms))
    dfac = torch.tensor(G / S0, dtype=torch.float32)
    sc_hidden = torch.tensor(1.0 / (G * nv), dtype=torch.float32)
    sc_last = torch.tensor(1.0 / (S0 * nv), dtype=torch.float32)
    grads = [None] * (2 * depth)
    delta = dlast
    link, inv = fp8_layer_scales(params, first_omega_0, hidden_omega_0)
    # round 3: at hidden 256 / depth >= 3 (k_fwd_pipe + k_bwd8h) layer 0's phases are spilled as bytes too and the backward
    # of layer 1 decodes them like any hidden layer's; elsewhere they are re-derived from the coordinates (fp32)
    l0_bytes = params[0].shape[0] == 256 and depth >= 3
    for l in range(L, 0, -1):
        if l - 1 == 0:
            ph = _phase_q8(t0) * (1.0 / 256.0) + PHASE_EPS if (l0_bytes and l != L) else ph0
        else:
            ph = q[l - 1] * (1.0 / 256.0) + PHASE_EPS
        act = _rt(torch.sin(TWO_PI * ph.double()).float(), "f16")
        sc = sc_last if l == L else sc_hidden * torch.tensor(inv[l], dtype=torch.float32)
        grads[2 * l] = (delta.t() @ act) * sc
        grads[2 * l + 1] = delta.sum(0) * sc
        om = first_omega_0 if l - 1 == 0 else hidden_omega_0
        Gacc = delta @ _rt(params[2 * l] * torch.tensor(om * link[l], dtype=torch.float32), "f16")   # (link: a power of two in the image)
        c = torch.cos(TWO_PI * ph.double()).float()
        if l == L:
            c = c * dfac
        delta = _e4m3(Gacc * c)
    xh = _rt(x, "f16")
    xl = _rt(x - xh, "f16")
    sc0 = sc_hidden * torch.tensor(inv[0], dtype=torch.float32)
    grads[0] = (delta.t() @ xh + delta.t() @ xl) * sc0
    grads[1] = delta.sum(0) * sc0
    return sse / (img.shape[-1] * float(n)), sse, grads, pred.reshape(h, w, -1)
