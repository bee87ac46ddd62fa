"""ctypes binding of libsiren_fit.so (C ABI: include/siren_fit.h).

PyTorch is used only for device memory, streams and (in parallel.py) torch.distributed; every
arithmetic step of the hot path runs in the HIP library.  Missing library => ImportError-like
RuntimeError at load time (no fallback path exists).
"""
import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIREN_FIT_LIB: developer override to A/B kernel variants built side by side (same C ABI)
_LIB_PATH = os.environ.get("SIREN_FIT_LIB") or os.path.normpath(os.path.join(_HERE, "..", "csrc", "libsiren_fit.so"))

SF_ABI_VERSION = 2
DTYPES = {"bf16": 0, "f16": 1}


class sf_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32),
        ("in_features", C.c_int32), ("out_features", C.c_int32), ("hidden", C.c_int32), ("depth", C.c_int32),
        ("first_omega_0", C.c_float), ("hidden_omega_0", C.c_float),
        ("outermost_linear", C.c_int32), ("compute_dtype", C.c_int32),
        ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("device", C.c_int32), ("stream", C.c_void_p), ("chunk_pixels", C.c_int64),
        ("scratch_format", C.c_int32),
    ]


_lib = None


def load_library():
    """Load libsiren_fit.so once and declare the prototypes of include/siren_fit.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} not found: build it with `python __graft_entry__.py build` (hipcc "
            "--offload-arch=gfx950). The SIREN engine has no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    H, F, I64 = C.c_void_p, C.c_void_p, C.c_int64  # F: device float*
    proto = {
        "sf_create": [C.POINTER(sf_config), C.POINTER(H)],
        "sf_destroy": [H],
        "sf_abi_version": [],
        "sf_num_params": [H, C.POINTER(I64)],
        "sf_scratch_format": [H, C.POINTER(C.c_int32)],
        "sf_param_offset": [H, C.c_int32, C.POINTER(I64), C.POINTER(I64)],
        "sf_set_params": [H, F], "sf_get_params": [H, F], "sf_set_masks": [H, F],
        "sf_get_grads": [H, F], "sf_set_grads": [H, F],
        "sf_get_adam_state": [H, F, F, C.POINTER(I64)], "sf_set_adam_state": [H, F, F, I64],
        "sf_state_ptr": [H, C.c_int32, C.POINTER(C.c_void_p)],
        "sf_sse_ptr": [H, C.POINTER(C.c_void_p)],
        "sf_debug_scratch": [H, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(I64)],
        "sf_debug_throw": [C.c_int32],
        "sf_kmeans_fit": [H, F, I64, F, C.c_int32, C.c_int32, C.c_float, F, C.c_int32, C.c_void_p, C.c_void_p, F],
        "sf_params_changed": [H],
        "sf_set_coords": [H, F, F], "sf_set_target": [H, F],
        "sf_forward": [H, F, C.POINTER(C.c_double)],
        "sf_forward_backward": [H, C.POINTER(C.c_double)],
        "sf_adam_step": [H, C.c_float],
        "sf_step": [H, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float)],
        "sf_profile_enable": [H, C.c_int32], "sf_profile_reset": [H], "sf_set_graph_replay": [H, C.c_int32],
        "sf_profile_num_kernels": [H, C.POINTER(C.c_int32)],
        "sf_profile_get": [H, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(I64),
                           C.POINTER(C.c_double), C.POINTER(C.c_double)],
    }
    for name, args in proto.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.sf_last_error.argtypes = []
    lib.sf_last_error.restype = C.c_char_p
    if lib.sf_abi_version() != SF_ABI_VERSION:
        raise RuntimeError("libsiren_fit.so ABI version mismatch")
    _lib = lib
    return lib


def exported_symbols() -> Sequence[str]:
    """Names include/siren_fit.h declares (used by the CPU test that the library exports them all)."""
    hdr = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "siren_fit.h"))
    import re
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(sf_[a-z_0-9]+)\s*\(", txt)))


def _check(rc: int):
    if rc != 0:
        raise RuntimeError(f"siren_fit error {rc}: {load_library().sf_last_error().decode()}")


class _DevView:
    """Zero-copy torch view of engine-owned device memory via __cuda_array_interface__."""

    def __init__(self, ptr: int, n: int, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _f32_cuda(t: torch.Tensor, n: Optional[int] = None) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("expected a contiguous float32 CUDA tensor")
    if n is not None and t.numel() != n:
        raise ValueError(f"expected {n} elements, got {t.numel()}")
    return t


class SirenEngine:
    """One per-image fit on one HIP stream (one sf_handle)."""

    STATE = {"params": 0, "grads": 1, "exp_avg": 2, "exp_avg_sq": 3, "masks": 4}

    def __init__(self, height: int, width: int, hidden: int, depth: int, first_omega_0: float = 50.0,
                 hidden_omega_0: float = 30.0, outermost_linear: bool = True, out_features: int = 3,
                 compute_dtype: str = "f16", device: int = 0, row_begin: int = 0, row_end: int = 0,
                 chunk_pixels: int = 0, betas=(0.9, 0.999), eps: float = 1e-8, scratch_format: int = 0):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("SirenEngine needs a gfx950 GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device("cuda", device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
        cfg = sf_config(SF_ABI_VERSION, height, width, row_begin, row_end, 2, out_features, hidden, depth,
                        first_omega_0, hidden_omega_0, int(bool(outermost_linear)), DTYPES[compute_dtype],
                        betas[0], betas[1], eps, device, stream, chunk_pixels, scratch_format)
        self.h = C.c_void_p()
        _check(self.lib.sf_create(C.byref(cfg), C.byref(self.h)))
        n = C.c_int64()
        _check(self.lib.sf_num_params(self.h, C.byref(n)))
        self.num_params = n.value
        self.height, self.width, self.hidden, self.depth = height, width, hidden, depth
        self.row_begin, self.row_end = row_begin, (row_end if row_end else height)
        self.npix = (self.row_end - self.row_begin) * width
        self.out_features = out_features
        self._target = None
        self._views = {}

    @property
    def scratch_format(self) -> int:
        """sf_config.scratch_format in use (what 0 / auto resolved to; an auto handle moves to 16 when a mask is set)"""
        f = C.c_int32()
        _check(self.lib.sf_scratch_format(self.h, C.byref(f)))
        return f.value

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.sf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ---------------------------------------------------------------------------
    def param_offsets(self, layer: int):
        w, b = C.c_int64(), C.c_int64()
        _check(self.lib.sf_param_offset(self.h, layer, C.byref(w), C.byref(b)))
        return w.value, b.value

    def view(self, which: str) -> torch.Tensor:
        """Flat fp32 torch view (no copy) of engine state: params | grads | exp_avg | exp_avg_sq | masks."""
        if which not in self._views:
            p = C.c_void_p()
            _check(self.lib.sf_state_ptr(self.h, self.STATE[which], C.byref(p)))
            self._views[which] = torch.as_tensor(_DevView(p.value, self.num_params), device=self.device)
        return self._views[which]

    def grad_view(self) -> torch.Tensor:
        """The engine's own flat fp32 gradient (zero-copy): what pixel-split ranks all-reduce in place."""
        return self.view("grads")

    def sse_view(self) -> torch.Tensor:
        """1-element float64 view of the device scalar the last pass wrote its sum of squared residuals to."""
        if "sse" not in self._views:
            p = C.c_void_p()
            _check(self.lib.sf_sse_ptr(self.h, C.byref(p)))
            self._views["sse"] = torch.as_tensor(_DevView(p.value, 1, "<f8"), device=self.device)
        return self._views["sse"]

    def debug_scratch(self, which: str) -> torch.Tensor:
        """uint8 view of an engine scratch tensor of the last pass: phases | deltas | dlast | slabs (tests only)."""
        p, n = C.c_void_p(), C.c_int64()
        _check(self.lib.sf_debug_scratch(self.h, {"phases": 0, "deltas": 1, "dlast": 2, "slabs": 3}[which], C.byref(p), C.byref(n)))
        return torch.as_tensor(_DevView(p.value, n.value, "|u1"), device=self.device)

    def kmeans_fit(self, weight: torch.Tensor, guess: torch.Tensor, iter_limit: int = 5, tol: float = 1e-4):
        """sf_kmeans_fit on this handle's stream, no host sync: (centroids [K+1, zero padded], n_centroids [device int32],
        labels [int64, weight's shape], new_weight)."""
        w = _f32_cuda(weight.reshape(-1))
        centers = _f32_cuda(guess.reshape(-1)).clone()
        K = centers.numel()
        cent = torch.empty(K + 1, device=self.device)
        ncent = torch.empty(1, dtype=torch.int32, device=self.device)
        labels = torch.empty(w.numel(), dtype=torch.int64, device=self.device)
        new_w = torch.empty_like(w)
        _check(self.lib.sf_kmeans_fit(self.h, w.data_ptr(), w.numel(), centers.data_ptr(), K, iter_limit, tol, cent.data_ptr(), K + 1,
                                      ncent.data_ptr(), labels.data_ptr(), new_w.data_ptr()))
        return cent, ncent, labels.reshape(weight.shape), new_w.reshape(weight.shape)

    def params_changed(self):
        _check(self.lib.sf_params_changed(self.h))

    def set_params(self, flat: torch.Tensor):
        _check(self.lib.sf_set_params(self.h, _f32_cuda(flat, self.num_params).data_ptr()))

    def get_params(self) -> torch.Tensor:
        out = torch.empty(self.num_params, device=self.device)
        _check(self.lib.sf_get_params(self.h, out.data_ptr()))
        return out

    def get_grads(self) -> torch.Tensor:
        out = torch.empty(self.num_params, device=self.device)
        _check(self.lib.sf_get_grads(self.h, out.data_ptr()))
        return out

    def set_grads(self, flat: torch.Tensor):
        _check(self.lib.sf_set_grads(self.h, _f32_cuda(flat, self.num_params).data_ptr()))

    def set_masks(self, flat: Optional[torch.Tensor]):
        _check(self.lib.sf_set_masks(self.h, None if flat is None else _f32_cuda(flat, self.num_params).data_ptr()))

    def get_adam_state(self):
        m = torch.empty(self.num_params, device=self.device)
        v = torch.empty(self.num_params, device=self.device)
        step = C.c_int64()
        _check(self.lib.sf_get_adam_state(self.h, m.data_ptr(), v.data_ptr(), C.byref(step)))
        return m, v, step.value

    def set_adam_state(self, m: torch.Tensor, v: torch.Tensor, step: int):
        _check(self.lib.sf_set_adam_state(self.h, _f32_cuda(m, self.num_params).data_ptr(),
                                          _f32_cuda(v, self.num_params).data_ptr(), step))

    # ---- data ----------------------------------------------------------------------------
    def set_coords(self, rows: torch.Tensor, cols: torch.Tensor):
        _check(self.lib.sf_set_coords(self.h, _f32_cuda(rows, self.height).data_ptr(),
                                      _f32_cuda(cols, self.width).data_ptr()))

    def set_target(self, img: torch.Tensor):
        self._target = _f32_cuda(img, self.npix * self.out_features)  # keep alive: the engine borrows it
        _check(self.lib.sf_set_target(self.h, self._target.data_ptr()))

    # ---- hot path ------------------------------------------------------------------------
    def forward(self, want_pred: bool = True, want_sse: bool = True):
        pred = torch.empty(self.row_end - self.row_begin, self.width, self.out_features,
                           device=self.device) if want_pred else None
        sse = C.c_double()
        _check(self.lib.sf_forward(self.h, pred.data_ptr() if want_pred else None,
                                   C.byref(sse) if want_sse else None))
        return pred, (sse.value if want_sse else None)

    def forward_backward(self, sync: bool = True) -> Optional[float]:
        sse = C.c_double()
        _check(self.lib.sf_forward_backward(self.h, C.byref(sse) if sync else None))
        return sse.value if sync else None

    def adam_step(self, lr: float):
        _check(self.lib.sf_adam_step(self.h, lr))

    def step(self, lrs: Sequence[float], want_loss: bool = False):
        n = len(lrs)
        arr = (C.c_float * n)(*lrs)
        out = (C.c_float * n)() if want_loss else None
        _check(self.lib.sf_step(self.h, arr, n, out))
        return list(out) if want_loss else None

    def set_graph_replay(self, on: bool):
        """step(): replay a captured hipGraph per training step instead of launching kernel by kernel"""
        _check(self.lib.sf_set_graph_replay(self.h, int(on)))

    # ---- measurement -----------------------------------------------------------------------
    def profile(self, on: bool):
        _check(self.lib.sf_profile_enable(self.h, int(on)))

    def profile_reset(self):
        _check(self.lib.sf_profile_reset(self.h))

    def profile_report(self):
        n = C.c_int32()
        _check(self.lib.sf_profile_num_kernels(self.h, C.byref(n)))
        rep = {}
        for i in range(n.value):
            name, ms, cnt, fl, by = C.c_char_p(), C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            _check(self.lib.sf_profile_get(self.h, i, C.byref(name), C.byref(ms), C.byref(cnt), C.byref(fl),
                                           C.byref(by)))
            rep[name.value.decode()] = {"total_ms": ms.value, "launches": cnt.value,
                                        "flops_per_launch": fl.value, "bytes_per_launch": by.value}
        return rep
