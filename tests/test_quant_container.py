"""'Next' rows §8f-1/2 on CPU: k-means quantisation and the compressed container against golden vectors
produced by the reference's own code (tests/golden/make_golden_quant.py)."""
import numpy as np
import pytest
import torch

from implicit_image.models import registry
from implicit_image.pipeline import entropy_coding
from implicit_image.pipeline.quant import KmeansQuant, find_centroids


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("layer", [1, 2])
def test_find_centroids_matches_reference(golden, bits, layer):
    d = golden("kmeans_64x64")
    w = torch.tensor(d[f"b{bits}_l{layer}_weight"])
    cent, labels, new_w = find_centroids(w, 2 ** bits)
    assert np.array_equal(labels.numpy(), d[f"b{bits}_l{layer}_labels"])            # index path: bit-exact
    assert np.array_equal(cent.numpy(), d[f"b{bits}_l{layer}_centroids"])           # same scatter_mean restatement
    assert np.array_equal(new_w.numpy(), d[f"b{bits}_l{layer}_new_weight"])
    assert (new_w[w == 0] == 0).all()                                               # pruned weights stay exactly zero


def _quantised_model(golden):
    hot = golden("hot_64x4_256")
    torch.manual_seed(0)
    model = registry["siren"](depth=4, hidden_size=64, first_omega_0=50, hidden_omega_0=30)
    off = 0
    with torch.no_grad():
        for p in model.parameters():
            n = p.numel()
            p.copy_(torch.tensor(hot["final"][off:off + n]).view(p.shape))
            off += n
        g = torch.Generator().manual_seed(11)
        model.layers[1].linear.weight.mul_((torch.rand(64, 64, generator=g) > 0.4).float())
    optim = torch.optim.Adam(model.parameters(), lr=3e-4)
    comp = KmeansQuant(model, optim, bits=8, skip_ll=["layers.0.linear", "layers.3.linear"])
    comp.kmeans_modify_weights()          # what the forward-pre-hooks do on the reference's first forward
    comp.update_weights()
    return model


@pytest.mark.parametrize("stream", ["plain", "lzma"])
def test_container_bytes_and_meta_match_reference(golden, tmp_path, stream):
    d = golden("container_64x4")
    model = _quantised_model(golden)
    assert list(model.state_dict().keys()) == [str(k) for k in d["state_dict_keys"]]
    half = model.half()
    size = entropy_coding.compress_state_dict(half, tmp_path, stream_name=stream)
    data = np.frombuffer(open(tmp_path / "compressed_weights.data", "rb").read(), np.uint8)
    assert np.array_equal(data, d[f"{stream}_bytes"])                               # byte-exact stream
    assert open(tmp_path / "meta_data.json").read() == str(d[f"{stream}_meta"])     # byte-exact metadata
    assert size == int(d[f"{stream}_reported_size"])
    dec = entropy_coding.decompress_state_dict(tmp_path, stream_name=stream)
    for k in dec:
        assert np.array_equal(dec[k].numpy(), d[f"decoded::{k}"])


def test_parser_roundtrip_like_the_reference_test(tmp_path):
    """The reference's only asserting test (entropy_coding/parsers.py:66-93): write -> read round trip."""
    arr = np.random.default_rng(0).random((3, 3))
    for cls in (entropy_coding.NumpyParser, entropy_coding.LZMAParser):
        f = open(tmp_path / "x.bin", "wb+")
        with cls(f) as p:
            n = p.write(arr)
            assert n > 0
            assert p.flush() == n and p._written == 0
        f.seek(0)
        with cls(f) as p:
            back = np.frombuffer(p.read(), dtype=arr.dtype).reshape(3, 3)
        assert np.array_equal(back, arr)
        f.close()
    with pytest.raises(NotImplementedError):
        entropy_coding.compress_state_dict(torch.nn.Linear(2, 2), tmp_path, stream_name="huffman")
