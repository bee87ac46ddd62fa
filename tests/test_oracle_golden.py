"""The CPU oracle (oracle/siren_oracle.py) against golden vectors minted from the REAL reference
(tests/golden/make_golden.py).  CPU only.  Float tolerances are stated per assertion."""
import hashlib

import numpy as np
import torch

from oracle import siren_oracle as so


def test_grid_matches_reference(golden):
    g = golden("grid")
    assert np.array_equal(so.get_grid(5, 7).numpy(), g["grid_5x7"])          # bit-exact
    assert np.array_equal(so.grid_vectors(4096, 256)[0].numpy(), g["lin4096"])
    assert np.array_equal(so.grid_vectors(4096, 256)[1].numpy(), g["lin256"])


def test_init_draw_order_bit_exact(golden):
    for name, hidden, depth in (("grads_64x4_32", 64, 4), ("grads_256x8_32", 256, 8), ("grads_128x6_48", 128, 6),
                                ("wide_512x3_32", 512, 3)):
        assert np.array_equal(so.flatten(so.siren_init(hidden, depth, seed=0)), golden(name)["init"])


def test_first_step_loss_and_gradients(golden):
    for name, hidden, depth in (("grads_64x4_32", 64, 4), ("grads_256x8_32", 256, 8), ("grads_128x6_48", 128, 6),
                                ("wide_512x3_32", 512, 3)):
        d = golden(name)
        p = so.unflatten(d["init"], hidden, depth)
        H, W, _ = d["img"].shape
        grid = so.get_grid(H, W)
        loss, _, grads = so.loss_and_grads(p, grid, torch.tensor(d["img"]))
        assert abs(loss - float(d["loss"])) <= 1e-6 * float(d["loss"])
        g = so.flatten(grads)
        assert np.linalg.norm(g - d["grads"]) <= 2e-6 * np.linalg.norm(d["grads"])   # fp32 summation-order noise
        assert np.abs(so.forward(p, grid).numpy() - d["pred"]).max() <= 2e-6


def test_sine_output_layer(golden):
    """outermost_linear=False: first-step loss / prediction / gradients of the oracle against the reference."""
    d = golden("sine_out_64x3_16")
    H, W, _ = d["img"].shape
    p = so.unflatten(d["init"], 64, 3)
    assert np.array_equal(so.flatten(so.siren_init(64, 3, seed=0)), d["init"])
    grid, img = so.get_grid(H, W), torch.tensor(d["img"])
    loss, _, grads = so.loss_and_grads(p, grid, img, outermost_linear=False)
    assert abs(loss - float(d["loss"])) <= 1e-6 * float(d["loss"])
    assert np.linalg.norm(so.flatten(grads) - d["grads"]) <= 2e-6 * np.linalg.norm(d["grads"])
    assert np.abs(so.forward(p, grid, outermost_linear=False).numpy() - d["pred"]).max() <= 2e-6


def test_wide_short_run(golden):
    """512x3 (wide path fixture): 10 Adam steps of the oracle against the reference's loss curve."""
    d = golden("wide_512x3_32")
    H, W, _ = d["img"].shape
    p = so.unflatten(d["init"], 512, 3)
    opt = so.Adam(p)
    grid, img = so.get_grid(H, W), torch.tensor(d["img"])
    losses = np.array([so.train_epoch(p, opt, grid, img, t) for t in range(10)])
    assert np.abs(losses / d["losses"] - 1).max() <= 1e-3


def test_config1_training_run(golden):
    """config 1 of BASELINE.json: SIREN 64x4, 256x256, 1000 full-batch Adam steps."""
    d = golden("hot_64x4_256")
    img = so.synthetic_image(256, 256)
    assert hashlib.sha256(img.numpy().tobytes()).hexdigest() == str(d["img_sha256"])
    grid = so.get_grid(256, 256)
    p = so.siren_init(64, 4, seed=0)
    assert np.array_equal(so.flatten(p), d["init"])
    opt = so.Adam(p)
    losses = np.array([so.train_epoch(p, opt, grid, img, t) for t in range(1000)])
    ref = d["losses"]
    # identical arithmetic up to fp32 summation order for the first ~100 steps; afterwards the
    # trajectories decorrelate at loss spikes (the reference itself does across thread counts)
    assert np.max(np.abs(losses[:100] - ref[:100]) / ref[:100]) <= 1e-5
    assert np.median(np.abs(losses - ref) / ref) <= 5e-3
    _, mse, psnr, psnr8 = so.eval_epoch(p, grid, img)
    assert abs(psnr - float(d["psnr"])) <= 0.05            # the north-star parity bar, dB
    assert abs(psnr8 - float(d["psnr8"])) <= 0.05


def test_short_256x8_run(golden):
    d = golden("short_256x8_64")
    p = so.unflatten(d["init"], 256, 8)
    grid, img = so.get_grid(64, 64), torch.tensor(d["img"])
    opt = so.Adam(p)
    losses = np.array([so.train_epoch(p, opt, grid, img, t) for t in range(20)])
    assert np.max(np.abs(losses - d["losses"]) / d["losses"]) <= 2e-4
    assert np.abs(so.flatten(p) - d["final"]).max() <= 2e-4
