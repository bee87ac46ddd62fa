#!/usr/bin/env python3
"""Golden vectors for the "next" rows of SURVEY.md §8(f): k-means weight quantisation and the
compressed-container wire format, minted by running the REAL reference code in this container.

    python tests/golden/make_golden_quant.py

Absent third-party modules are replaced by inert stubs, except ONE arithmetic stand-in:
`torch_scatter.scatter_mean` (absent, unpinned in requirements.txt:28-29) is restated from its documented
semantics (sum / clamp(count, 1), output size index.max()+1).  Everything that flows through it — the
k-means centroids — is therefore PARITY UNPINNED at that boundary; labels, container
layout and byte streams are the reference's own code.
Writes: kmeans_64x64.npz, container_64x4.npz (plain + lzma byte streams and meta_data.json).
"""
import importlib.util
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def scatter_mean(src, index, dim=0):
    """torch_scatter.scatter_mean semantics for dim=0 and a 1-D index over rows of `src`."""
    n = int(index.max().item()) + 1
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    out.index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype))
    return out / cnt.clamp(min=1).reshape(-1, *([1] * (src.dim() - 1)))


class _TT:
    def __class_getitem__(cls, item):
        return cls


def main():
    sys.path.insert(0, REF)
    _stub("omegaconf", DictConfig=dict, OmegaConf=object)
    _stub("torch_optimizer", Shampoo=object)
    _stub("torch_scatter", scatter_mean=scatter_mean)
    _stub("torchtyping", TensorType=_TT)
    _stub("zstandard")
    from implicit_image.pipeline.quant.kmeans import KmeansQuant
    from implicit_image.pipeline import entropy_coding
    spec = importlib.util.spec_from_file_location("ref_siren", f"{REF}/implicit_image/models/siren.py")
    siren = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(siren)

    hot = np.load(f"{OUT}/hot_64x4_256.npz")
    torch.manual_seed(0)
    model = siren.Siren(depth=4, hidden_size=64, first_omega_0=50, hidden_omega_0=30)
    # load the trained config-1 weights and zero ~40 % of layer 1 (a sparse layer exercises the non-zero rule)
    off = 0
    with torch.no_grad():
        for p in model.parameters():
            n = p.numel()
            p.copy_(torch.tensor(hot["final"][off:off + n]).view(p.shape))
            off += n
        g = torch.Generator().manual_seed(11)
        model.layers[1].linear.weight.mul_((torch.rand(64, 64, generator=g) > 0.4).float())
    optim = torch.optim.Adam(model.parameters(), lr=3e-4)
    out = {}
    for bits in (4, 8):
        comp = KmeansQuant(model, optim, bits=bits, skip_ll=["layers.0.linear", "layers.3.linear"])
        for li in (1, 2):
            lin = model.layers[li].linear
            w = lin.weight.data.clone()
            cent, labels, new_w = comp.find_centroids(lin)
            out[f"b{bits}_l{li}_weight"] = w.numpy()
            out[f"b{bits}_l{li}_centroids"] = cent.numpy()
            out[f"b{bits}_l{li}_labels"] = labels.numpy().astype(np.int16)
            out[f"b{bits}_l{li}_new_weight"] = new_w.numpy()
        for h in [*comp.forward_pre_hook_ll, *comp.backward_hook_ll]:
            h.remove()
    np.savez_compressed(f"{OUT}/kmeans_64x64.npz", **out)

    # ---- round 3: a 256 x 256 layer (the metric model's hidden size), 30 % of it pruned, bits 5 (the reference's default,
    # conf/quant/kmeans.yaml) and 8 ---------------------------------------------------------------------------------------
    torch.manual_seed(0)
    big = siren.Siren(depth=4, hidden_size=256, first_omega_0=50, hidden_omega_0=30)
    with torch.no_grad():
        g = torch.Generator().manual_seed(12)
        big.layers[1].linear.weight.mul_((torch.rand(256, 256, generator=g) > 0.3).float())
    out2 = {"weight": big.layers[1].linear.weight.data.clone().numpy()}
    for bits in (5, 8):
        comp = KmeansQuant(big, torch.optim.Adam(big.parameters(), lr=3e-4), bits=bits, skip_ll=["layers.0.linear", "layers.3.linear"])
        cent, labels, new_w = comp.find_centroids(big.layers[1].linear)
        out2[f"b{bits}_centroids"] = cent.numpy()
        out2[f"b{bits}_labels"] = labels.numpy().astype(np.int16)
        for h in [*comp.forward_pre_hook_ll, *comp.backward_hook_ll]:
            h.remove()
    np.savez_compressed(f"{OUT}/kmeans_256x256.npz", **out2)
    print("kmeans 256x256: centroids", {k: v.shape for k, v in out2.items() if "centroids" in k})
    if os.environ.get("KMEANS_ONLY"):
        return
    print("kmeans: centroids", {k: v.shape for k, v in out.items() if "centroids" in k})

    # ---- container: quantise (8 bit), convert, half(), compress with plain and lzma --------------
    comp = KmeansQuant(model, optim, bits=8, skip_ll=["layers.0.linear", "layers.3.linear"])
    model(torch.rand(3, 3, 2, generator=torch.Generator().manual_seed(1)))      # fires the forward pre-hooks
    comp.update_weights()
    sd_keys = list(model.state_dict().keys())
    half = model.half()
    res = {"state_dict_keys": np.array(sd_keys)}
    for stream in ("plain", "lzma"):
        with tempfile.TemporaryDirectory() as d:
            nbytes = entropy_coding.compress_state_dict(half, d, stream_name=stream)
            res[f"{stream}_bytes"] = np.frombuffer(open(os.path.join(d, "compressed_weights.data"), "rb").read(), np.uint8)
            res[f"{stream}_meta"] = np.array(open(os.path.join(d, "meta_data.json")).read())
            res[f"{stream}_reported_size"] = nbytes
            dec = entropy_coding.decompress_state_dict(d, stream_name=stream) if stream == "plain" else None
            if dec is not None:
                for k, v in dec.items():
                    res[f"decoded::{k}"] = v.numpy()
    for k, v in half.state_dict().items():
        res[f"half::{k}"] = v.float().numpy() if v.dtype.is_floating_point else v.numpy()
    np.savez_compressed(f"{OUT}/container_64x4.npz", **res)
    print("container: plain", res["plain_bytes"].size, "lzma", res["lzma_bytes"].size, "keys", sd_keys)


if __name__ == "__main__":
    main()
