"""Other masking modes on CPU: one `update_connections()` replayed from the reference's
captured pre-update state (weights, gradients, Adam moments, masks, decay position) must give the
reference's post-update masks / weights / moments bit-exactly (index paths).  Golden: make_golden_masking.py."""
import numpy as np
import pytest
import torch

from implicit_image.models import registry
from implicit_image.pipeline.masking import LinearDecay, MagnitudePruneDecay
from implicit_image.utils.train_helper import setup_mask


class Cfg(dict):
    __getattr__ = dict.get


MODES = {
    "snfs": Cfg(name="SNFS", density=0.3, sparse_init="erdos-renyi-kernel", dense_gradients=True,
                growth_mode="momentum", prune_mode="magnitude", redistribution_mode="momentum", dense=False,
                prune_rate=0.1, decay_schedule="cosine", end_when=60, interval=5),
    "pruning": Cfg(name="Pruning", density=1.0, sparse_init="random", final_density=0.5, dense_gradients=True,
                   growth_mode="none", prune_mode="global-magnitude", redistribution_mode="none", dense=False,
                   decay_schedule="magnitude-prune", start_when=5, end_when=60, interval=5),
    "set": Cfg(name="SET", density=0.5, sparse_init="erdos-renyi-kernel", dense_gradients=False,
               growth_mode="random", prune_mode="magnitude", redistribution_mode="none", dense=False,
               prune_rate=0.2, decay_schedule="linear", end_when=60, interval=5),
}


def _bits(mask, names):
    return np.packbits(np.concatenate([mask.mask_dict[n].cpu().numpy().ravel().astype(np.uint8) for n in names]))


def _load_flat(params, flat, attr=None, optim=None, key=None):
    off = 0
    for p in params:
        n = p.numel()
        t = torch.tensor(flat[off:off + n]).view(p.shape)
        if optim is not None:
            optim.state[p][key] = t
        elif attr == "grad":
            p.grad = t
        else:
            p.data = t
        off += n


@pytest.mark.parametrize("tag", ["snfs", "pruning", "set"])
@pytest.mark.parametrize("upd", [5, 10])
def test_update_connections_bit_exact(golden, tag, upd):
    d = golden(f"masking_{tag}")
    mcfg = MODES[tag]
    torch.manual_seed(0)
    m = registry["siren"](depth=4, hidden_size=64, first_omega_0=50, hidden_omega_0=30)
    optim = torch.optim.Adam(m._param_list(), lr=3e-4)          # plain torch Adam: the masking code only reads .state
    mask = setup_mask(m, optim, mcfg)
    names = [str(n) for n in d["mask_names"]]
    assert [n for n, _ in m.named_parameters() if n in mask.mask_dict] == names
    assert np.array_equal(_bits(mask, names), d["mask_init"])   # init scheme incl. 'random' (first layer dropped)
    pre, post = f"u{upd}_", f"a{upd}_"
    params = m._param_list()
    _load_flat(params, d[pre + "w"])
    _load_flat(params, d[pre + "g"], attr="grad")
    for p in params:
        optim.state[p]["step"] = torch.tensor(float(upd + 1))
    _load_flat(params, d[pre + "m"], optim=optim, key="exp_avg")
    _load_flat(params, d[pre + "v"], optim=optim, key="exp_avg_sq")
    mb = np.unpackbits(d[pre + "mask"])
    off = 0
    for n in names:
        sh = mask.mask_dict[n].shape
        k = int(np.prod(sh))
        mask.mask_dict[n] = torch.tensor(mb[off:off + k].astype(np.float32)).view(sh)
        off += k
    mask.mask_step = int(d[pre + "mask_step"])
    mask.adjusted_growth = float(d[pre + "adjusted_growth"])
    mask.adjustments = list(d[pre + "adjustments"])
    mask.prune_threshold = float(d[pre + "prune_threshold"])
    mask.stats.total_nonzero, mask.stats.total_zero = int(d[pre + "total_nonzero"]), int(d[pre + "total_zero"])
    # bring the decay object to the reference's position
    dec = mask.prune_rate_decay
    if tag == "pruning":
        dec.current_prune_rate = float(d[pre + "rate"])
    else:
        for s_ in range(mask.mask_step):
            dec.step(s_)
    assert mask.prune_rate == pytest.approx(float(d[pre + "rate"]), rel=0, abs=1e-15)
    torch.manual_seed(1000 + upd)                                # pins random growth like the generator script
    mask.update_connections()
    assert np.array_equal(_bits(mask, names), d[post + "mask"])
    assert [int(mask.mask_dict[n].sum().item()) for n in names] == d[post + "nnz"].tolist()
    got_w = np.concatenate([p.detach().numpy().ravel() for p in params])
    assert np.array_equal(got_w, d[post + "w"])
    got_m = np.concatenate([optim.state[p]["exp_avg"].numpy().ravel() for p in params])
    assert np.array_equal(got_m, d[post + "m"])                  # reset_momentum when dense_gradients=False
    assert mask.mask_step == int(d[post + "mask_step"])
    assert mask.adjusted_growth == pytest.approx(float(d[post + "adjusted_growth"]), rel=1e-12, abs=1e-12)
    assert mask.prune_threshold == pytest.approx(float(d[post + "prune_threshold"]), rel=1e-12)
    assert mask.stats.total_density == pytest.approx(float(d[post + "density"]), rel=0, abs=1e-15)


def test_linear_and_magnitude_prune_decay(golden):
    d = golden("decay_linear_magprune")
    lin = LinearDecay(prune_rate=0.2, T_max=20)
    seq = []
    for s in range(30):
        lin.step(s)
        seq.append(lin.get_dr())
    assert np.allclose(seq, d["linear"], rtol=0, atol=1e-16)
    mp = MagnitudePruneDecay(final_sparsity=0.5, T_max=60, T_start=5, interval=5)
    seq = []
    for s in range(70):
        mp.step(s, 0.01 * s)
        seq.append(mp.get_dr())
    assert np.allclose(seq, d["magprune"], rtol=0, atol=1e-16)
