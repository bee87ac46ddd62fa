"""Compressed-model container ("next" row §8f-2; reference: pipeline/entropy_coding/__init__.py:15-186,
parsers.py:20-64).

Wire format (unchanged): `compressed_weights.data` = the state-dict tensors in state-dict order, each
as raw little-endian ndarray.tobytes() pushed through the stream writer; quantised layers contribute
`centroids` (float) + `labeled_weight` (uint8 labels) instead of `weight`; `meta_data.json` maps the
write order to {shape, dtype, name} (sorted keys, indent 2).  Writers: plain, lzma (one independent LZMA
blob per tensor; the reported size counts sys.getsizeof of every blob, parsers.py:56), zstd only when the
`zstandard` module is importable (it is not, offline).  `huffman` is unimplemented in the reference."""
import json
import lzma
import sys
from collections import OrderedDict
from pathlib import Path
from typing import Dict, Union

import numpy as np
import torch
from torch import nn


class NumpyParser:
    def __init__(self, handler):
        self.handler, self._written = handler, 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def read(self, *a, **k):
        return self.handler.read(*a, **k)

    def write(self, array: np.ndarray) -> int:
        self._written += self.handler.write(array.tobytes())
        return self._written

    def flush(self):
        self.handler.flush()
        n, self._written = self._written, 0
        return n


class LZMAParser(NumpyParser):
    def read(self, **kwargs):
        return lzma.decompress(self.handler.read(**kwargs))

    def write(self, array: np.ndarray) -> int:
        blob = lzma.compress(array.tobytes())
        self._written += sys.getsizeof(blob)
        self.handler.write(blob)
        return self._written


def _stream(name: str, write: bool, **kwargs):
    if name == "plain":
        return NumpyParser
    if name == "lzma":
        return LZMAParser
    if name == "zstd":
        try:
            import zstandard
        except ImportError as e:
            raise NotImplementedError("entropy_coding=zstd needs the `zstandard` module (not installed)") from e
        return (zstandard.ZstdCompressor(level=kwargs["level"]).stream_writer if write
                else zstandard.ZstdDecompressor().stream_reader)
    raise NotImplementedError(f"stream writer {name} not found.")


def linear_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    """state_dict on the CPU; quantised Linear layers drop `.weight` and cast labels to uint8
    (reference __init__.py:15-41)."""
    state_dict = OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items())
    for name, module in model.named_modules():
        if isinstance(module, nn.Linear) and hasattr(module, "centroids") and hasattr(module, "labeled_weight"):
            state_dict.pop(f"{name}.weight")
            if f"{name}.labeled_weight" not in state_dict:
                raise KeyError("Please run .update() before compressing weights")
            labels = state_dict[f"{name}.labeled_weight"]
            if labels.max() > 2 ** 8:
                raise NotImplementedError("more than 256 centroids: the reference asks for torch.uint16 here")
            state_dict[f"{name}.labeled_weight"] = labels.to(torch.uint8)
    return state_dict


def compress_state_dict(model: nn.Module, dir_name: Union[str, Path], stream_name: str, **kwargs) -> int:
    writer = _stream(stream_name, True, **kwargs)
    state_dict = linear_state_dict(model)
    dir_name = Path(dir_name)
    dir_name.mkdir(exist_ok=True, parents=True)
    meta = OrderedDict()
    with open(dir_name / "compressed_weights.data", "wb") as fh:
        with writer(fh) as comp:
            for order, (name, tensor) in enumerate(state_dict.items()):
                array = tensor.numpy()
                comp.write(array)
                meta[order] = {"shape": array.shape, "dtype": str(array.dtype), "name": name}
            comp.flush()
    with open(dir_name / "meta_data.json", "w") as f:
        f.write(json.dumps(meta, indent=2, sort_keys=True))
    return (dir_name / "compressed_weights.data").stat().st_size


def decompress_state_dict(dir_name: Union[str, Path], stream_name: str, **kwargs) -> Dict[str, torch.Tensor]:
    """Inverse (reference __init__.py:123-186); codebook layers come back as dense `.weight`.
    (For lzma the reference's reader cannot decode its own multi-blob stream; here every blob is decoded.)"""
    dir_name = Path(dir_name)
    with open(dir_name / "meta_data.json") as f:
        meta = {int(k): v for k, v in json.load(f).items()}
    raw = open(dir_name / "compressed_weights.data", "rb").read()
    if stream_name == "lzma":
        dec, rest = b"", raw
        while rest:
            d = lzma.LZMADecompressor()
            dec += d.decompress(rest)
            rest = d.unused_data
    elif stream_name == "plain":
        dec = raw
    else:
        import io
        with _stream(stream_name, False, **kwargs)(io.BytesIO(raw)) as r:
            dec = r.read()
    arrays, offset = {}, 0
    for order in sorted(meta):
        shape, dtype, name = meta[order]["shape"], np.dtype(meta[order]["dtype"]), meta[order]["name"]
        count = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        arrays[name] = np.frombuffer(dec, dtype=dtype, count=count, offset=offset).reshape(shape)
        offset += count * dtype.itemsize
    out = {}
    for name, arr in arrays.items():
        if "centroids" not in name and "labeled_weight" not in name:
            out[name] = torch.from_numpy(arr.copy()).float()
        elif "labeled_weight" in name:
            cent = arrays[name.replace("labeled_weight", "centroids")]
            out[name.replace("labeled_weight", "weight")] = torch.from_numpy(cent[arr].copy()).float()
    return out
