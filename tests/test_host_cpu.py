"""CPU tests of the host side: conf/ loader grammar, image loading, row/job sharding, and the
pixel-split all-reduce path with world_size 2 over gloo (the oracle stands in for the engine)."""
import os

import numpy as np
import pytest
import torch

from implicit_image.config import expand_sweeps, load_config
from implicit_image.data import get_grid, grid_vectors, load_img, read_ppm, synthetic_image
from implicit_image.parallel import PixelSplitFit, shard_jobs, shard_rows
from oracle import siren_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = os.path.join(ROOT, "conf")


def test_config_defaults_groups_overrides_interpolation():
    c = load_config(CONF, [])
    assert c.mlp.name == "siren" and c.mlp.hidden_size == 128 and c.mlp.depth == 8      # conf/mlp/siren.yaml
    assert c.optim.lr == pytest.approx(3e-4) and isinstance(c.optim.lr, float)
    assert c.masking.name == "RigL" and c.quant.name == "KMeans" and c.entropy_coding.stream_name == "plain"   # the reference's default composition
    d = load_config(CONF, ["masking=none", "quant=none"])
    assert not d.masking and not d.quant                                                # dense fit
    assert c.exp_name == "siren_synthetic" and c.train.batch_height == c.img.height
    c = load_config(CONF, ["masking=RigL", "masking.density=0.1", "+mlp.hidden_size=256", "img=flower"], cwd="/x")
    assert c.masking.name == "RigL" and c.masking.density == 0.1 and c.masking.prune_rate == 0.1
    assert c.mlp.hidden_size == 256
    assert c.img.path == "/x/img/rgb16bit/flower_foveon.ppm"                           # ${cwd}, ${img.bits}, ${img.name}
    with pytest.raises(ValueError):
        load_config(CONF, ["masking=NoSuch"])
    assert expand_sweeps(["a=1,2", "b=x"]) == [["a=1", "b=x"], ["a=2", "b=x"]]


def test_grid_and_vectors_roundtrip():
    g = get_grid(5, 7)
    assert torch.equal(g, so.get_grid(5, 7))
    r, c = grid_vectors(g)
    assert torch.equal(r, torch.linspace(0, 1, 5)) and torch.equal(c, torch.linspace(0, 1, 7))
    with pytest.raises(ValueError):
        grid_vectors(torch.rand(5, 7, 2))


def test_load_img_synthetic_and_ppm(tmp_path):
    assert torch.equal(load_img("synthetic", 32, 40), so.synthetic_image(32, 40))
    assert torch.equal(synthetic_image(16, 16, seed=7), so.synthetic_image(16, 16, seed=7))
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 65536, size=(6, 10, 3), dtype=np.uint16)
    p = tmp_path / "t.ppm"
    with open(p, "wb") as f:
        f.write(b"P6\n# comment\n10 6\n65535\n")
        f.write(arr.astype(">u2").tobytes())
    raw = read_ppm(str(p))
    assert np.array_equal(raw.numpy(), arr.astype(np.int32))
    img = load_img(str(p), height=4, width=8, bits=16, crop_mode="centre-crop")
    ref = torch.from_numpy((arr.astype(np.float64) / 65535).astype(np.float32))[1:5, 1:9]
    assert torch.equal(img, ref)


def test_sharding_helpers():
    assert [shard_rows(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_rows(4096, 8, r) for r in (0, 7)] == [(0, 512), (3584, 4096)]
    assert shard_jobs(list(range(10)), 4, 1) == [1, 5, 9]


class _OracleShard:
    """Stand-in for a row-sharded SirenEngine: the fp32 oracle on rows [r0, r1) of the image."""

    def __init__(self, params, grid, img, r0, r1):
        self.p, self.grid, self.img, self.n_total = params, grid[r0:r1], img[r0:r1], grid.shape[0] * grid.shape[1]
        self.opt, self.g = so.Adam(params), None

    def forward_backward(self, sync=False):
        _, sse, grads = so.loss_and_grads(self.p, self.grid, self.img, n_total=self.n_total)
        self.g = torch.tensor(so.flatten(grads))
        self.sse = torch.tensor([sse], dtype=torch.float64)

    def grad_view(self):
        return self.g

    def sse_view(self):
        return self.sse

    def adam_step(self, lr):
        self.opt.step(self.p, so.unflatten(self.g.numpy(), 32, 3), lr=lr)


def _pixel_split_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    H, W = 12, 9
    p = so.siren_init(32, 3, seed=0)
    grid, img = so.get_grid(H, W), so.synthetic_image(H, W, seed=5)
    r0, r1 = shard_rows(H, world, rank)
    fit = PixelSplitFit(_OracleShard(p, grid, img, r0, r1), 3 * H * W)
    losses = [fit.step(3e-4) for _ in range(5)]
    q.put((rank, losses, so.flatten(p)))
    dist.barrier()
    dist.destroy_process_group()


def test_pixel_split_gloo_world2_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pixel_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    out = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    # single-process full-image reference
    H, W = 12, 9
    p = so.siren_init(32, 3, seed=0)
    grid, img = so.get_grid(H, W), so.synthetic_image(H, W, seed=5)
    opt = so.Adam(p)
    ref = [so.train_epoch(p, opt, grid, img, t) for t in range(5)]
    for rank, losses, params in out:
        assert np.allclose(losses, ref, rtol=1e-5, atol=0)
        assert np.allclose(params, so.flatten(p), rtol=0, atol=2e-6)
    assert np.array_equal(out[0][2], out[1][2])            # replicas stay bit-identical
