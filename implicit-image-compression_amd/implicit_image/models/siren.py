"""SIREN model with the reference's constructor, parameter names and call signature
(reference: implicit_image/models/siren.py:9-134), executed by the gfx950 engine.

The torch module only OWNS NAMES AND SHAPES: once bound, every `layers.{i}.linear.{weight,bias}`
Parameter (and its .grad) is a zero-copy view of the engine's flat fp32 state, so optimiser /
masking / quantisation code that pokes `weight.data`, `weight.grad` or iterates
`named_parameters()` sees live engine state.  forward() runs the fused HIP kernels; there is no
PyTorch arithmetic fallback.
"""
import copy
import math
from typing import Optional

import numpy as np
import torch
from torch import nn

from ..data import grid_vectors


class SineLayer(nn.Module):
    """Linear -> sin(omega_0 * z) (reference siren.py:9-68).  Holds the nn.Linear so that
    `isinstance(m, nn.Linear)` scans (masking, k-means, entropy coding) find it under `.linear`."""

    def __init__(self, in_features: int, out_features: int, has_bias: bool = True, is_first: bool = False,
                 omega_0: float = 30.0, no_activation: bool = False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.is_first, self.omega_0, self.no_activation = is_first, omega_0, no_activation
        # nn.Linear's default init draws (weight, then bias) come first, as in the reference, so the
        # generator state and the bias values match a reference model built from the same seed
        self.linear = nn.Linear(in_features, out_features, bias=has_bias)
        bound = 1 / in_features if is_first else np.sqrt(6 / in_features) / omega_0
        with torch.no_grad():
            self.linear.weight.uniform_(-bound, bound)
        self.linear.scaler = bound

    def forward(self, x):  # pragma: no cover - the layers never run individually
        raise RuntimeError("SineLayer is executed by the fused engine; call Siren.forward(grid)")


class Siren(nn.Module):
    def __init__(self, input_size: int = 2, output_size: int = 3, depth: int = 8, hidden_size: int = 128,
                 first_omega_0: float = 50.0, hidden_omega_0: float = 50.0, outermost_linear: bool = True,
                 simulate_quantization: bool = False, small_dense_density: float = 1.0,
                 compute_dtype: str = "f16", chunk_pixels: int = 0, scratch_format: int = 0, **kwargs):
        super().__init__()
        if simulate_quantization:
            raise NotImplementedError("simulate_quantization (QAT stubs) is outside the accelerated path")
        hidden_size = int(hidden_size * np.sqrt(small_dense_density))   # Small_Dense (reference siren.py:88)
        layers = [SineLayer(input_size, hidden_size, is_first=True, omega_0=first_omega_0)]
        for _ in range(depth - 2):
            layers.append(SineLayer(hidden_size, hidden_size, omega_0=hidden_omega_0))
        layers.append(SineLayer(hidden_size, output_size, omega_0=hidden_omega_0, no_activation=outermost_linear))
        self.layers = nn.Sequential(*layers)
        self.simulate_quantization = simulate_quantization
        self.cfg = dict(input_size=input_size, output_size=output_size, depth=depth, hidden_size=hidden_size,
                        first_omega_0=float(first_omega_0), hidden_omega_0=float(hidden_omega_0),
                        outermost_linear=bool(outermost_linear), compute_dtype=compute_dtype,
                        chunk_pixels=chunk_pixels, scratch_format=int(scratch_format))
        # callbacks run right before every engine pass / right after every backward: the seam the reference
        # fills with per-Linear forward-pre and backward hooks (k-means quantisation, pipeline/quant/kmeans.py:39-55)
        self.pre_pass_callbacks = []
        self.post_backward_callbacks = []
        # hidden widths the kernels are instantiated for (<= 256: fused chain kernels; 512 / 1024: layer-at-a-time
        # kernels); any other width (e.g. Small_Dense's int(hidden * sqrt(density)), reference siren.py:88) runs
        # zero-padded to the next one: padded neurons have zero weights and bias, output sin(0) = 0 and receive
        # exactly zero gradients, so they stay zero
        self._engine_width = next((w for w in (32, 64, 128, 256, 512, 1024) if w >= hidden_size), None)
        if self._engine_width is None:
            raise NotImplementedError(f"hidden_size {hidden_size} > 1024 is not supported by the gfx950 engine")
        if self._engine_width > 256 and depth < 3:
            raise NotImplementedError("hidden_size > 256 needs depth >= 3")
        self._padded = self._engine_width != hidden_size
        self._adam = ((0.9, 0.999), 1e-8)      # torch.optim.Adam defaults; EngineAdam overrides (conf/optim/*.yaml)
        self._pad_index = None
        self._engine = None
        self._engine_key = None
        self._grid_key = None
        self._target_key = None

    # ---- engine binding -------------------------------------------------------------------
    def set_scratch_format(self, fmt: int):
        """sf_config.scratch_format of the engine (0 auto / 8 / 12 / 16); a live engine of another format is rebuilt on
        the next pass: engine() carries the parameters, the Adam moments and step count (sf_get/set_adam_state) and the
        masks over to the new handle and rebinds every EngineAdam created on this model, so a switch in the middle of a
        fit neither resets the optimiser nor leaves `optimizer.state[p]` pointing at freed device memory."""
        if self.cfg["scratch_format"] != int(fmt):
            self.cfg["scratch_format"] = int(fmt)

    def set_adam_hparams(self, betas, eps: float):
        """Adam betas / eps of the engine's fused optimiser kernel (sf_config); a live engine created with other
        values is rebuilt on the next pass (its moments restart, as with a new torch optimiser)."""
        self._adam = (tuple(betas), float(eps))

    def _param_list(self):
        """The engine's parameters in flat order: (weight, bias) of every layer.  Not `self.parameters()`:
        k-means conversion registers extra (frozen) `centroids` / `labeled_weight` parameters on the Linears."""
        out = []
        for layer in self.layers:
            out += [layer.linear.weight, layer.linear.bias]
        return out

    def engine(self, grid: torch.Tensor, img: Optional[torch.Tensor] = None, row_begin: int = 0, row_end: int = 0,
               full_height: Optional[int] = None):
        """Engine bound to this model for `grid` (created on first use / when the image size changes)."""
        from .._engine import SirenEngine
        if not grid.is_cuda:
            raise RuntimeError("Siren runs on the gfx950 engine only: move model, grid and image to 'cuda'")
        h, w, _ = grid.shape
        H = full_height or h
        key = (H, w, row_begin, row_end, grid.device.index, self._adam, self.cfg["scratch_format"])
        if self._engine is None or self._engine_key != key:
            c = self.cfg
            carry = None
            if self._engine is not None:
                old = self._engine
                m, v, st = old.get_adam_state()
                masks = old.view("masks").clone() if getattr(self, "_has_engine_mask", False) else None
                carry = (m, v, st, masks, old.num_params, (old.height, old.width, old.row_begin, old.row_end))
                self._unbind()
            self._engine = SirenEngine(H, w, self._engine_width, c["depth"], c["first_omega_0"], c["hidden_omega_0"],
                                       c["outermost_linear"], c["output_size"], c["compute_dtype"],
                                       device=grid.device.index or 0, row_begin=row_begin, row_end=row_end,
                                       chunk_pixels=c["chunk_pixels"], betas=self._adam[0], eps=self._adam[1],
                                       scratch_format=c["scratch_format"])
            self._engine_key, self._grid_key, self._target_key = key, None, None
            new = self._engine
            if carry is not None and carry[4] == new.num_params and carry[5] == (new.height, new.width, new.row_begin, new.row_end):
                # same fit on a re-created handle (another scratch format / Adam hyper-parameters): the optimiser goes along
                new.set_adam_state(carry[0], carry[1], carry[2])
                if carry[3] is not None:
                    new.set_masks(carry[3])
            else:
                self._has_engine_mask = False
            for opt in list(getattr(self, "_engine_optims", ())):
                opt._bound = None
                if not self._padded:
                    self._sync_to_engine()
                    opt._bind_state(new)
        eng = self._engine
        gkey = (grid.data_ptr(), tuple(grid.shape))
        if self._grid_key != gkey:
            rows, cols = grid_vectors(grid)
            if full_height and full_height != h:
                raise ValueError("pass the full-height grid in pixel-split mode")
            eng.set_coords(rows.float(), cols.float())
            self._grid_key = gkey
        if img is not None:
            tkey = (img.data_ptr(), tuple(img.shape), img._version)
            if self._target_key != tkey:
                eng.set_target(img.contiguous().float())
                self._target_key = tkey
        for cb in list(self.pre_pass_callbacks):
            cb()
        self._sync_to_engine()
        return eng

    def _padded_index(self, device):
        """flat index of every logical parameter element inside the engine's (wider) flat layout"""
        if self._pad_index is None or self._pad_index.device != device:
            c, wp = self.cfg, self._engine_width
            idx, off = [], 0
            for l in range(c["depth"]):
                fin = c["input_size"] if l == 0 else c["hidden_size"]
                fout = c["output_size"] if l == c["depth"] - 1 else c["hidden_size"]
                fin_p = c["input_size"] if l == 0 else wp
                fout_p = c["output_size"] if l == c["depth"] - 1 else wp
                r = torch.arange(fout, device=device)[:, None] * fin_p + torch.arange(fin, device=device)[None, :]
                idx.append((off + r).reshape(-1))
                off += fin_p * fout_p
                idx.append(off + torch.arange(fout, device=device))
                off += fout_p
            self._pad_index = torch.cat(idx)
        return self._pad_index

    def _sync_to_engine(self):
        """(Re)bind every Parameter to its slice of the engine's flat buffers.  Code that REPLACED
        `weight.data` (e.g. `weight.data = weight.data * mask`) is detected by pointer and copied in.
        Padded widths: parameters stay ordinary tensors and are scattered into the engine every pass."""
        eng = self._engine
        if self._padded:
            flat = torch.zeros(eng.num_params, device=eng.device)
            flat[self._padded_index(eng.device)] = torch.cat([p.data.reshape(-1).float() for p in self._param_list()])
            eng.set_params(flat)
            return
        flat, grads = eng.view("params"), eng.view("grads")
        off = 0
        for p in self._param_list():
            n = p.numel()
            dst = flat[off:off + n].view(p.shape)
            if p.data.data_ptr() != dst.data_ptr():
                dst.copy_(p.data.to(dst.dtype))
                p.data = dst
            g = grads[off:off + n].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g
            off += n
        eng.params_changed()   # in-place edits through the views are invisible to the engine: always refresh

    def _gather_from_engine(self, which: str):
        """padded widths only: logical slices of an engine state vector ('params' | 'grads' | 'exp_avg' | ...)"""
        flat = self._engine.view(which)[self._padded_index(self._engine.device)]
        out, off = [], 0
        for p in self._param_list():
            out.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return out

    def download_grads(self):
        if self._padded:
            for p, g in zip(self._param_list(), self._gather_from_engine("grads")):
                p.grad = g.clone()

    def download_params(self):
        if self._padded:
            with torch.no_grad():
                for p, v in zip(self._param_list(), self._gather_from_engine("params")):
                    p.data.copy_(v)

    def set_engine_masks(self, flat_logical: torch.Tensor):
        """0/1 mask per logical parameter element -> engine (scattered into the wider layout when padded)."""
        eng = self._engine
        self._has_engine_mask = True
        if self._padded:
            full = torch.zeros(eng.num_params, device=eng.device)
            full[self._padded_index(eng.device)] = flat_logical.to(eng.device).float()
            eng.set_masks(full)
        else:
            eng.set_masks(flat_logical.contiguous())

    def _unbind(self):
        for p in self._param_list():
            p.data = p.data.clone()
            p.grad = None if not self._padded else p.grad
        self._engine.close()
        self._engine = None

    def __deepcopy__(self, memo):
        c = self.cfg
        new = Siren(c["input_size"], c["output_size"], c["depth"], c["hidden_size"], c["first_omega_0"],
                    c["hidden_omega_0"], c["outermost_linear"], compute_dtype=c["compute_dtype"],
                    chunk_pixels=c["chunk_pixels"], scratch_format=c["scratch_format"])
        new.to(next(self.parameters()).device)
        new._adam = self._adam
        with torch.no_grad():
            for a, b in zip(new._param_list(), self._param_list()):
                a.copy_(b)
        new.train(self.training)
        return new

    def half(self):
        """model.half() of the reference's save path (compress.py:246-250): detaches from the engine."""
        if self._engine is not None:
            self._unbind()
        return super().half()

    # ---- reference call signature -----------------------------------------------------------
    def forward(self, grid: torch.Tensor) -> torch.Tensor:
        """[H, W, 2] grid -> [H, W, output_size] prediction in [0, 1] (reference siren.py:123-134)."""
        pred, _ = self.engine(grid).forward(want_pred=True, want_sse=False)
        return pred
