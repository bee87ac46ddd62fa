"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP engine, called through the C ABI
(ctypes stub in implicit_image/_engine.py), against the CPU oracle on the same seeded inputs and
against golden vectors minted from the real reference.

Tolerances (all stated where used):
  forward, fp16 operands   : |dpred| <= 3e-4 abs        (operand rounding 2^-11, f32 accumulate)
  forward, bf16 operands   : |dpred| <= 3e-3 abs        (2^-8)
  gradients, fp16 operands, scratch_format 16 (unorm16 phases, fp16 deltas):
                             per-layer and total L2 error <= 5e-3 relative (measured ~1.3e-3)
  gradients, scratch_format 12 (phase BYTES; the default) and 8 (+ fp8 deltas): the quantisation is zero-mean noise
                             that averages out over the pixels a gradient sums (measured at 256x8: 1.0e-2 on 1 280
                             pixels, 3.0e-3 on 16 384, format 12; 3.4e-2 / 1.0e-2, format 8).  The bound is the noise
                             the numerics model (oracle/engine_model.py, same rounding points on the CPU) shows
                             against the fp32 reference on the same input: engine error <= 1.5 x model error + 2e-3,
                             and the engine must equal that model to <= 3e-3 (format 12; format 8 only at depth <= 4:
                             fp8 roundings decorrelate under 1e-6 perturbations, layer by layer)
  gradients, bf16 operands : <= 3e-2 relative (measured ~9e-3; compute_dtype="bf16" option)
  PSNR after equal steps   : |dPSNR| <= 0.05 dB          (BASELINE.json north star)
  index / mask paths       : bit-exact
"""
import math

import numpy as np
import pytest
import torch

from oracle import siren_oracle as so

pytestmark = pytest.mark.gpu


FORMATS = (16, 12, 8)      # sf_config.scratch_format; 0 / auto (fp16 operands, hidden <= 256) = 8 from 2^20 pixels, 12 below


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))


def _engine(H, W, hidden, depth, dtype="f16", params=None, img=None, **kw):
    from implicit_image._engine import SirenEngine
    eng = SirenEngine(H, W, hidden, depth, compute_dtype=dtype, **kw)
    gh, gw = so.grid_vectors(H, W)
    eng.set_coords(gh.cuda(), gw.cuda())
    if params is not None:
        eng.set_params(torch.tensor(so.flatten(params)).cuda())
    if img is not None:
        r0, r1 = eng.row_begin, eng.row_end
        eng.set_target(img[r0:r1].contiguous().cuda())
    return eng


@pytest.mark.parametrize("name,hidden,depth", [("grads_64x4_32", 64, 4), ("grads_256x8_32", 256, 8),
                                               ("grads_128x6_48", 128, 6)])
@pytest.mark.parametrize("dtype,fmt,tol,gtol", [("f16", 16, 3e-4, 5e-3), ("f16", 12, 3e-4, None), ("f16", 8, 3e-4, None),
                                                ("bf16", 16, 3e-3, 3e-2)])
def test_forward_and_gradients_vs_reference_golden(golden, name, hidden, depth, dtype, fmt, tol, gtol):
    from oracle import engine_model as em
    d = golden(name)
    H, W, _ = d["img"].shape
    p = so.unflatten(d["init"], hidden, depth)
    eng = _engine(H, W, hidden, depth, dtype, p, torch.tensor(d["img"]), scratch_format=fmt)
    pred, sse = eng.forward()
    assert np.abs(pred.cpu().numpy() - d["pred"]).max() <= tol
    assert abs(sse / (3 * H * W) - float(d["loss"])) <= 50 * tol * float(d["loss"])
    eng.forward_backward()
    g = eng.get_grads().cpu().numpy()
    ref = d["grads"]
    model = None
    if gtol is None:      # byte formats: bound = the numerics model's own quantisation noise on this input
        model = so.flatten(em.loss_and_grads(p, so.get_grid(H, W), torch.tensor(d["img"]), scratch=fmt)[2])
    bound = lambda a, b, m: gtol if gtol is not None else 1.5 * _rel(m, b) + 2e-3
    assert _rel(g, ref) <= bound(g, ref, model)
    off = 0
    for fin, fout in so.layer_dims(hidden, depth):
        for n in (fin * fout, fout):
            a, b = g[off:off + n], ref[off:off + n]
            assert _rel(a, b) <= bound(a, b, None if model is None else model[off:off + n])
            off += n


@pytest.mark.parametrize("H,W,hidden,depth", [(5, 7, 32, 3), (1, 1, 64, 2), (17, 300, 64, 4), (33, 31, 128, 5)])
def test_ragged_and_tiny_grids(H, W, hidden, depth):
    """Pixel counts that are not multiples of the 256-pixel workgroup tile, down to one pixel."""
    p = so.siren_init(hidden, depth, seed=1)
    img = so.synthetic_image(H, W, seed=2)
    grid = so.get_grid(H, W)
    loss, sse, grads = so.loss_and_grads(p, grid, img)
    eng = _engine(H, W, hidden, depth, "f16", p, img)
    pred, sse_e = eng.forward()
    assert (pred.cpu() - so.forward(p, grid)).abs().max().item() <= 3e-4
    assert abs(sse_e - sse) <= 2e-3 * sse
    eng.forward_backward()
    g, ref = eng.get_grads().cpu().numpy(), so.flatten(grads)
    from oracle import engine_model as em
    model = so.flatten(em.loss_and_grads(p, grid, img, scratch=12)[2])      # default format: phase bytes
    if not np.isfinite(model).all():     # 1x1: the model's fixed 2^20 delta pre-scale overflows fp16 (the engine scales by N)
        assert _rel(g, ref) <= 5e-3
        return
    assert _rel(g, model) <= 3e-3
    assert _rel(g, ref) <= 1.5 * _rel(model, ref) + 2e-3


@pytest.mark.parametrize("H,W,depth,dtype,fmt,chunk", [
    (19, 23, 3, "f16", 12, 0),       # one hidden layer: the odd-count tail of the layer loop, output layer on the second buffer
    (40, 33, 5, "f16", 12, 0),       # three hidden layers: one pair + the odd tail; 1 320 pixels = 5.2 groups of 256
    (31, 64, 7, "f16", 16, 0),       # five hidden layers, unorm16 phases (two stores per tile epilogue)
    (48, 50, 4, "bf16", 16, 0),      # even count, bf16 operands (layer 0 stays an fp16 split product)
    (300, 301, 6, "f16", 12, 32768), # 353 groups in three chunks: more groups than CUs in a chunk, ragged last group, per-chunk partials
    (64, 64, 8, "f16", 8, 1024),     # fp8 deltas: the chunk scale is formed from the persistent kernel's (fewer) SSE partials
])
def test_pipeline_forward_kernel_code_paths(H, W, depth, dtype, fmt, chunk):
    """k_fwd_pipe (hidden 256, depth >= 3) beyond the 256x8 / 256x6 shapes of the golden fixtures: odd and single hidden
    layer counts (its layer loop is unrolled by two with a separate tail), group counts that are not a multiple of the
    persistent grid, several chunks, every scratch format, bf16, and the evaluation form - against the fp32 oracle and the
    engine's numerics model."""
    from oracle import engine_model as em
    p = so.siren_init(256, depth, seed=3)
    img = so.synthetic_image(H, W, seed=4)
    grid = so.get_grid(H, W)
    loss, sse, grads = so.loss_and_grads(p, grid, img)
    eng = _engine(H, W, 256, depth, dtype, p, img, scratch_format=fmt, chunk_pixels=chunk)
    pred, sse_e = eng.forward()                                  # evaluation form (no phase stores)
    tol = 3e-4 if dtype == "f16" else 4e-3
    assert (pred.cpu() - so.forward(p, grid)).abs().max().item() <= tol
    assert abs(sse_e - sse) <= (2e-3 if dtype == "f16" else 2e-2) * sse
    sse_t = eng.forward_backward()                               # training form
    assert abs(sse_t - sse_e) <= 1e-6 * sse_e                    # same forward, other partial grouping
    g, ref = eng.get_grads().cpu().numpy(), so.flatten(grads)
    model = so.flatten(em.loss_and_grads(p, grid, img, fwd=dtype, scratch=fmt)[2])
    assert _rel(g, ref) <= 1.5 * _rel(model, ref) + (2e-3 if dtype == "f16" else 2e-2)
    if fmt != 8:                                                 # fp8 roundings decorrelate engine and model with depth
        assert _rel(g, model) <= (4e-3 if dtype == "f16" else 1.5e-2)
    g2 = eng.get_grads().clone()
    eng.forward_backward()
    assert torch.equal(eng.get_grads(), g2)                      # run-to-run bit-identical


def test_auto_scratch_format_rule_and_mask_switch():
    """sf_config.scratch_format = 0: phase bytes + fp8 deltas from 2^20 pixels, phase bytes + 16-bit deltas below, 16 for
    bf16 operands and wide layers; an auto handle that receives a mask moves to format 16 (and then equals a handle created
    with 16 bit for bit), an explicit format stays."""
    assert _engine(64, 64, 64, 4).scratch_format == 12
    assert _engine(1024, 1024, 256, 3).scratch_format == 8
    assert _engine(1023, 1024, 256, 3).scratch_format == 12
    assert _engine(64, 64, 64, 4, "bf16").scratch_format == 16
    assert _engine(32, 32, 512, 3).scratch_format == 16
    H, W, hidden, depth = 48, 40, 128, 5
    p = so.siren_init(hidden, depth, seed=5)
    img = so.synthetic_image(H, W, seed=6)
    gen = torch.Generator().manual_seed(2)
    flat = torch.cat([((torch.rand(q.shape, generator=gen) < 0.3).float() if q.dim() == 2 else torch.ones_like(q)).reshape(-1) for q in p])
    auto, fixed16, fixed12 = (_engine(H, W, hidden, depth, "f16", p, img, scratch_format=f) for f in (0, 16, 12))
    for e in (auto, fixed16, fixed12):
        e.set_masks(flat.cuda())
    assert (auto.scratch_format, fixed16.scratch_format, fixed12.scratch_format) == (16, 16, 12)
    la, lb = auto.step([3e-4] * 5, want_loss=True), fixed16.step([3e-4] * 5, want_loss=True)
    assert la == lb and torch.equal(auto.get_params(), fixed16.get_params())


def test_scratch_formats_agree_at_a_megapixel():
    """The auto format switches to fp8 deltas at 2^20 pixels.  The reference cannot run a megapixel 256x8 fit in test time,
    so the statement checked here is relative: on a 1024x1024 image the three scratch formats end an annealed 160-step fit
    within 0.02 dB of each other (format 16 itself is pinned to the reference at the plateau fixture), i.e. the fp8 rounding of
    the deltas - zero-mean, summed over a million pixels per gradient - does not move the fit."""
    H = W = 1024
    p = so.siren_init(256, 8, seed=0)
    img = so.synthetic_image(H, W, seed=3)
    lrs = [3e-4 * 0.5 ** (t // 40) for t in range(160)]
    psnr = {}
    for fmt in (16, 12, 0):
        eng = _engine(H, W, 256, 8, "f16", p, img, scratch_format=fmt)
        if fmt == 0:
            assert eng.scratch_format == 8
        eng.step(lrs)
        _, sse = eng.forward(want_pred=False)
        psnr[eng.scratch_format] = 10 * math.log10(3 * H * W / sse)
        del eng
    assert abs(psnr[8] - psnr[16]) <= 0.02 and abs(psnr[12] - psnr[16]) <= 0.02, psnr


@pytest.mark.parametrize("fmt", FORMATS)
def test_chunking_is_a_summation_order_change_only(fmt):
    """Formats 16 / 12: every stored value is a function of its own pixel, so chunking only reorders fp32 sums.
    Format 8 picks its power-of-two delta scale per chunk (from that chunk's residual): roundings differ, by the
    fp8 noise itself."""
    H, W, hidden, depth = 48, 56, 128, 6
    p = so.siren_init(hidden, depth, seed=0)
    img = so.synthetic_image(H, W, seed=7)
    a = _engine(H, W, hidden, depth, "f16", p, img, scratch_format=fmt)
    b = _engine(H, W, hidden, depth, "f16", p, img, chunk_pixels=512, scratch_format=fmt)
    sa, sb = a.forward_backward(), b.forward_backward()
    assert abs(sa - sb) <= 1e-6 * sa
    ga, gb = a.get_grads(), b.get_grads()
    assert (ga - gb).norm().item() <= (1e-5 if fmt != 8 else 2e-2) * ga.norm().item()


def test_run_to_run_determinism():
    """No float atomics anywhere: two fits from the same state are bit-identical."""
    H, W, hidden, depth = 64, 64, 64, 4
    p = so.siren_init(hidden, depth, seed=0)
    img = so.synthetic_image(H, W, seed=3)
    outs = []
    for _ in range(2):
        eng = _engine(H, W, hidden, depth, "f16", p, img)
        losses = eng.step([3e-4] * 25, want_loss=True)
        outs.append((losses, eng.get_params().cpu().numpy()))
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1])


def test_adam_kernel_matches_torch_op_order():
    hidden, depth = 64, 4
    p = so.siren_init(hidden, depth, seed=0)
    eng = _engine(8, 8, hidden, depth, "f16", p)
    opt = so.Adam(p)
    gen = torch.Generator().manual_seed(5)
    for t in range(3):
        grads = [torch.randn(q.shape, generator=gen) * 1e-3 for q in p]
        eng.set_grads(torch.tensor(so.flatten(grads)).cuda())
        eng.adam_step(3e-4)
        opt.step(p, grads, lr=3e-4)
    got = eng.get_params().cpu().numpy()
    assert np.abs(got - so.flatten(p)).max() <= 1e-7           # fp32, same op order
    m, v, step = eng.get_adam_state()
    assert step == 3
    assert np.abs(m.cpu().numpy() - so.flatten(opt.m)).max() <= 1e-9


def test_loss_curve_tracks_oracle_then_psnr_parity_config1(golden):
    """BASELINE.json config 1: SIREN 64x4 on the 256x256 formula image, 1000 full-batch steps,
    against the loss curve and PSNR the REAL reference produced (tests/golden/hot_64x4_256.npz)."""
    d = golden("hot_64x4_256")
    H = W = 256
    img = so.synthetic_image(H, W)
    p = so.unflatten(d["init"], 64, 4)
    eng = _engine(H, W, 64, 4, "f16", p, img)
    lrs = [so.step_lr(3e-4, t) for t in range(1000)]
    losses = np.array(eng.step(lrs, want_loss=True))
    ref = d["losses"]
    assert np.max(np.abs(losses[:50] - ref[:50]) / ref[:50]) <= 2e-3       # before trajectories decorrelate
    assert np.median(np.abs(losses - ref) / ref) <= 2e-2
    _, sse = eng.forward(want_pred=False)
    psnr = 10 * math.log10(3 * H * W / sse)
    assert abs(psnr - float(d["psnr"])) <= 0.05, (psnr, float(d["psnr"]))


def test_short_run_256x8_vs_reference(golden):
    """First steps of the metric model (256x8) on a 64x64 image against the real reference's loss curve
    and against the engine NUMERICS MODEL (oracle/engine_model.py: same rounding points, torch CPU).
    This heavily over-parameterised early phase is chaotic (an fp32 run with another thread count
    drifts by percents within 100 steps), so only the first steps are compared."""
    from oracle import engine_model as em
    d = golden("short_256x8_64")
    p = so.unflatten(d["init"], 256, 8)
    img = torch.tensor(d["img"])
    eng = _engine(64, 64, 256, 8, "f16", p, img)
    losses = np.array(eng.step([3e-4] * 20, want_loss=True))
    assert np.max(np.abs(losses[:4] - d["losses"][:4]) / d["losses"][:4]) <= 2e-3
    grid, opt, model = so.get_grid(64, 64), so.Adam(p), []
    for t in range(4):
        loss, _, grads, _ = em.loss_and_grads(p, grid, img, scratch=12)
        opt.step(p, grads, lr=3e-4)
        model.append(loss)
    # this fixture amplifies perturbations ~5x per step (measured: engine vs its own numerics model 2e-7, 8e-5, 2e-4,
    # 6e-4 .. 1.0e-3 at steps 0..3 depending on last-bit details of layer 0, 3e-2 at step 6), so the fourth step gets
    # the bound of the reference curve and later steps carry no parity information
    rel = np.abs(losses[:4] - np.array(model)) / np.array(model)
    assert np.max(rel[:3]) <= 5e-4 and rel[3] <= 2e-3, (losses, model)


@pytest.mark.parametrize("name,hidden,depth", [("grads_64x4_32", 64, 4), ("grads_256x8_32", 256, 8)])
@pytest.mark.parametrize("dtype,fmt", [("f16", 16), ("f16", 12), ("f16", 8), ("bf16", 16)])
def test_engine_equals_its_numerics_model(golden, name, hidden, depth, dtype, fmt):
    """Engine vs oracle/engine_model.py (identical rounding points): what remains is fp32 summation
    order, v_sin/v_cos vs libm, and the occasional rounding tie flipping: <= 2e-3 relative (fp16, format 16; measured
    5e-4), <= 3e-3 (bf16, whose 8-bit roundings flip 8x as hard; measured 2.4e-3), <= 4e-3 (format 12: a phase byte
    that flips moves one sine by up to 2.5e-2; measured 2.9e-3 at 256x8 on 1 280 pixels).  At hidden = 256 layer 0
    runs on the matrix pipe as a split-fp16 product (k_fwd_pipe, kL0Split): as accurate as the model's fp32 FMAs
    (6e-7 vs 8e-7 revolutions against float64) but not bit-identical to them, which costs a few more flips than the
    VALU layer 0 of the other widths (measured with SIREN_FIT_FWD_PIPE=0: 1.7e-3 / 4e-4 / 2.3e-3).  Format 8 is held to the model only at depth 4: an fp8 rounding that flips is a 6 % change of that
    delta, which flips more roundings in the next layer - at depth 8 engine and model decorrelate to the level of the
    fp8 noise itself (measured 2.3e-2, = model vs fp32), which test_forward_and_gradients_vs_reference_golden bounds."""
    from oracle import engine_model as em
    if fmt == 8 and depth > 4:
        pytest.skip("fp8 delta roundings decorrelate with depth: bounded against the fp32 reference instead")
    d = golden(name)
    H, W, _ = d["img"].shape
    p = so.unflatten(d["init"], hidden, depth)
    img = torch.tensor(d["img"])
    loss, sse, grads, pred = em.loss_and_grads(p, so.get_grid(H, W), img, fwd=dtype, scratch=fmt)
    eng = _engine(H, W, hidden, depth, dtype, p, img, scratch_format=fmt)
    pe, sse_e = eng.forward()
    assert (pe.cpu() - pred).abs().max().item() <= (5e-5 if dtype == "f16" else 4e-4)
    assert abs(sse_e - sse) <= 1e-4 * sse
    eng.forward_backward()
    g, ref = eng.get_grads().cpu().numpy(), so.flatten(grads)
    assert _rel(g, ref) <= (4e-3 if fmt != 16 else 3e-3 if dtype == "bf16" else 2e-3)


def test_masks_are_applied_inside_the_step():
    H, W, hidden, depth = 32, 32, 64, 4
    p = so.siren_init(hidden, depth, seed=0)
    img = so.synthetic_image(H, W, seed=9)
    eng = _engine(H, W, hidden, depth, "f16", p, img)
    gen = torch.Generator().manual_seed(1)
    masks, flat = [], []
    for q in p:
        mk = (torch.rand(q.shape, generator=gen) < 0.5).float() if q.dim() == 2 else torch.ones_like(q)
        masks.append(mk if q.dim() == 2 else None)
        flat.append(mk.reshape(-1))
        q.mul_(mk)
    eng.set_params(torch.tensor(so.flatten(p)).cuda())
    eng.set_masks(torch.cat(flat).cuda())
    opt = so.Adam(p)
    grid = so.get_grid(H, W)
    ref = [so.train_epoch(p, opt, grid, img, t, masks=masks) for t in range(10)]
    got = eng.step([3e-4] * 10, want_loss=True)
    assert np.max(np.abs(np.array(got) - np.array(ref)) / np.array(ref)) <= 2e-3
    w = eng.get_params().cpu()
    assert torch.all(w[torch.cat(flat) == 0] == 0)               # bit-exact: masked weights stay zero


@pytest.mark.parametrize("fmt", FORMATS)
def test_pixel_split_handles_compose(fmt):
    """Row-sharded handles (pixel-split mode): SSE and gradients of the shards add up to the
    full-image values — the property the RCCL all-reduce path relies on (format 8: up to its per-shard delta
    scale, i.e. to the fp8 noise)."""
    H, W, hidden, depth = 64, 48, 64, 4
    p = so.siren_init(hidden, depth, seed=0)
    img = so.synthetic_image(H, W, seed=4)
    full = _engine(H, W, hidden, depth, "f16", p, img, scratch_format=fmt)
    sse = full.forward_backward()
    g = full.get_grads()
    tot, gs = 0.0, torch.zeros_like(g)
    for r0, r1 in ((0, 20), (20, 64)):
        part = _engine(H, W, hidden, depth, "f16", p, img, row_begin=r0, row_end=r1, scratch_format=fmt)
        tot += part.forward_backward()
        gs += part.get_grads()
    assert abs(tot - sse) <= 1e-6 * sse
    assert (gs - g).norm().item() <= (1e-5 if fmt != 8 else 2e-2) * g.norm().item()


def test_full_size_grid_properties():
    """BASELINE metric shape (256x8) on a 2048x2048 grid: properties that need no oracle run —
    finite loss, gradient of a doubled residual doubles (linearity of the backward in dL/dout),
    loss decreases over a few steps."""
    H = W = 2048
    p = so.siren_init(256, 8, seed=0)
    ys = torch.linspace(0, 1, H, device="cuda")[:, None, None]
    xs = torch.linspace(0, 1, W, device="cuda")[None, :, None]
    k = torch.tensor([1.0, 2.0, 3.0], device="cuda")
    img = (0.5 + 0.25 * torch.sin(12 * xs * k) + 0.25 * torch.cos(9 * ys * k)).contiguous()
    eng = _engine(H, W, 256, 8, "f16", p)
    eng.set_target(img)
    pred, sse0 = eng.forward()
    assert math.isfinite(sse0) and pred.min().item() > -2 and pred.max().item() < 3
    eng.forward_backward()
    g1 = eng.get_grads().clone()
    img2 = (2 * img - pred).contiguous()          # residual (pred - img2) = 2 * (pred - img) ... negated twice
    eng.set_target(img2)
    eng.forward_backward()
    g2 = eng.get_grads()
    assert (g2 - 2 * g1).norm().item() <= 2e-2 * (2 * g1).norm().item()   # bf16 rounding of dL/dout differs
    eng.set_target(img)
    losses = eng.step([3e-4] * 5, want_loss=True)
    assert losses[-1] < losses[0]


def test_host_mirror_train_and_eval_epoch():
    """Reference-shaped API: registry['siren'] + get_optimizer_lr_scheduler + train_epoch/eval_epoch."""
    from implicit_image.data import get_grid
    from implicit_image.models import registry
    from implicit_image.utils.train_helper import eval_epoch, get_optimizer_lr_scheduler, train_epoch
    H = W = 64
    torch.manual_seed(0)
    model = registry["siren"](depth=4, hidden_size=64, first_omega_0=50, hidden_omega_0=30).to("cuda")
    img = so.synthetic_image(H, W, seed=3)
    grid = get_grid(H, W)
    optim, sched = get_optimizer_lr_scheduler(model, dict(name="adam", lr=3e-4))
    p = so.siren_init(64, 4, seed=0)
    opt = so.Adam(p)
    got = [train_epoch(model, optim, grid.cuda(), img.cuda(), lr_scheduler=sched) for _ in range(10)]
    ref = [so.train_epoch(p, opt, so.get_grid(H, W), img, t) for t in range(10)]
    assert np.max(np.abs(np.array(got) - np.array(ref)) / np.array(ref)) <= 2e-3
    pred, loss, psnr, psnr8 = eval_epoch(model, grid.cuda(), img.cuda())
    _, l_ref, psnr_ref, psnr8_ref = so.eval_epoch(p, so.get_grid(H, W), img)
    assert abs(psnr - psnr_ref) <= 0.05 and abs(psnr8 - psnr8_ref) <= 0.1
    # parameters are live views of engine state
    w0 = dict(model.named_parameters())["layers.1.linear.weight"]
    assert w0.grad is not None and w0.grad.shape == w0.shape
    assert optim.state[w0]["exp_avg"].abs().sum().item() > 0


def test_rigl_run_vs_reference_golden(golden):
    """RigL (ERK 0.5, prune 0.1, cosine) 120 steps on 64x64: same initial masks bit-exactly; the
    topology then evolves from bf16-rounded gradients, so the end state is compared by PSNR/density."""
    from implicit_image.data import get_grid
    from implicit_image.models import registry
    from implicit_image.utils.train_helper import (eval_epoch, get_optimizer_lr_scheduler, setup_mask, train_epoch)
    d = golden("rigl_64x4_64")

    class Cfg(dict):
        __getattr__ = dict.get
    H = W = 64
    torch.manual_seed(0)
    model = registry["siren"](depth=4, hidden_size=64, first_omega_0=50, hidden_omega_0=30).to("cuda")
    optim, sched = get_optimizer_lr_scheduler(model, Cfg(name="adam", lr=3e-4))
    mcfg = Cfg(name="RigL", density=0.5, sparse_init="erdos-renyi-kernel", dense_gradients=True,
               growth_mode="absolute-gradient", prune_mode="magnitude", redistribution_mode="none", dense=False,
               prune_rate=0.1, decay_schedule="cosine", end_when=90, interval=10)
    mask = setup_mask(model, optim, mcfg)
    names = [n for n, _ in model.named_parameters() if n in mask.mask_dict]
    bits = np.packbits(np.concatenate([mask.mask_dict[n].cpu().numpy().ravel().astype(np.uint8) for n in names]))
    assert np.array_equal(bits, d["mask0"])
    img, grid = torch.tensor(d["img"]).cuda(), get_grid(H, W).cuda()
    losses, rates = [], []
    for i in range(120):
        losses.append(train_epoch(model, optim, grid, img, lr_scheduler=sched, mask=mask))
        if i <= mcfg.end_when and i % mcfg.interval == 0:
            mask.update_connections()
        rates.append(mask.prune_rate)
    assert np.allclose(rates, d["rates"], rtol=0, atol=1e-12)            # schedule is exact
    assert mask.mask_step == int(d["mask_step"])
    assert np.max(np.abs(np.array(losses[:10]) - d["losses"][:10]) / d["losses"][:10]) <= 3e-3
    assert abs(mask.stats.total_density - float(d["density"][-1])) <= 0.01
    _, _, psnr, _ = eval_epoch(model, grid, img)
    assert abs(psnr - float(d["psnr"])) <= 0.3     # different (equally valid) topology after 10 updates
    for n, w in model.named_parameters():
        if n in mask.mask_dict:
            assert torch.all(w.data[mask.mask_dict[n] == 0] == 0)


def test_make_fit_entry_runs_the_reference_loop(tmp_path, monkeypatch):
    """`make fit` path: conf/ loader -> registry model -> train/eval loop of compress.py:137-170."""
    import os
    from implicit_image.config import load_config
    from implicit_image.fit import fit_one
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(tmp_path)
    cfg = load_config(os.path.join(root, "conf"), ["img.height=64", "img.width=64", "img.seed=3", "mlp.hidden_size=64",
                                                   "mlp.depth=4", "train.num_steps=100", "train.log_steps=50", "masking=none", "quant=none"])
    res = fit_one(cfg, torch.device("cuda", 0), str(tmp_path / "out"))
    p = so.siren_init(64, 4, seed=0)
    img, grid, opt = so.synthetic_image(64, 64, seed=3), so.get_grid(64, 64), so.Adam(p)
    for t in range(100):
        so.train_epoch(p, opt, grid, img, t)
    _, _, psnr_ref, _ = so.eval_epoch(p, grid, img)
    assert abs(res["PSNR"] - psnr_ref) <= 0.05
    sd = torch.load(tmp_path / "out" / "model.pth", weights_only=True)["state_dict"]
    assert list(sd)[:2] == ["layers.0.linear.weight", "layers.0.linear.bias"]


def test_make_fit_quant_and_compress_tail(tmp_path, monkeypatch):
    """compress.py:172-263 on the engine: k-means (8 bit) fine-tune, convert, half(), plain container;
    the container decodes back to the codebook weights and the quantised model keeps the PSNR."""
    import os
    from implicit_image.config import load_config
    from implicit_image.fit import fit_one
    from implicit_image.pipeline import entropy_coding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(tmp_path)
    cfg = load_config(os.path.join(root, "conf"), ["img.height=64", "img.width=64", "mlp.hidden_size=64", "mlp.depth=4",
                                                   "train.num_steps=300", "train.log_steps=300", "masking=none", "quant=kmeans",
                                                   "quant.num_steps=10", "quant.log_steps=10"])
    res = fit_one(cfg, torch.device("cuda", 0), str(tmp_path / "out"))
    # the 10-step fine-tune re-clusters every forward and is chaotic (27..30 dB seen for a 32.6 dB model); the
    # index path itself is pinned bit-exactly on the CPU (tests/test_quant_container.py), this is a flow check
    assert 20.0 < res["Quant PSNR"] <= res["PSNR"] + 0.5
    meta_dir = tmp_path / "out" / "model_quantized"
    dec = entropy_coding.decompress_state_dict(meta_dir, "plain")
    assert set(dec) == {f"layers.{i}.linear.{k}" for i in range(4) for k in ("weight", "bias")}
    assert res["Compressed Bytes"] == os.path.getsize(meta_dir / "compressed_weights.data")
    # 2 quantised 64x64 layers: 4096 one-byte labels each instead of 8192 bytes of fp16
    assert res["Compressed Bytes"] < 2 * (64 * 64 * 1 + 256 * 2) + (64 * 2 + 64 + 3 * 64 + 3 + 128) * 2 + 64


def test_small_dense_width_runs_zero_padded():
    """masking=Small_Dense narrows the net to int(hidden*sqrt(density)) (reference siren.py:88): 57 for 128 @ 0.2.
    The engine runs it zero-padded to 64; losses must track the fp32 oracle at the LOGICAL width."""
    from implicit_image.data import get_grid
    from implicit_image.models import registry
    from implicit_image.utils.train_helper import eval_epoch, get_optimizer_lr_scheduler, train_epoch
    H = W = 48
    torch.manual_seed(0)
    model = registry["siren"](depth=4, hidden_size=128, first_omega_0=50, hidden_omega_0=30,
                              small_dense_density=0.2).to("cuda")
    assert model.cfg["hidden_size"] == 57 and model._engine_width == 64
    p = [q.detach().cpu().clone() for q in model._param_list()]
    img, grid = so.synthetic_image(H, W, seed=3), get_grid(H, W)
    optim, sched = get_optimizer_lr_scheduler(model, dict(name="adam", lr=3e-4))
    opt = so.Adam(p)
    got = [train_epoch(model, optim, grid.cuda(), img.cuda(), lr_scheduler=sched) for _ in range(30)]
    ref = [so.train_epoch(p, opt, so.get_grid(H, W), img, t) for t in range(30)]
    assert np.max(np.abs(np.array(got) - np.array(ref)) / np.array(ref)) <= 2e-3
    w1 = dict(model.named_parameters())["layers.1.linear.weight"]
    assert tuple(w1.shape) == (57, 57) and w1.grad is not None and tuple(w1.grad.shape) == (57, 57)
    assert (w1.detach().cpu() - p[2]).abs().max().item() <= 2e-3      # 30 Adam steps of 3e-4 each; sign flips on ~0 gradients
    _, _, psnr, _ = eval_epoch(model, grid.cuda(), img.cuda())
    _, _, psnr_ref, _ = so.eval_epoch(p, so.get_grid(H, W), img)
    assert abs(psnr - psnr_ref) <= 0.05


# ---- wide path (hidden 512 / 1024: layer-at-a-time kernels, BASELINE configs 3 and 5) -------------------
@pytest.mark.parametrize("dtype,tol,gtol", [("f16", 3e-4, 5e-3), ("bf16", 3e-3, 3e-2)])
def test_wide_512_forward_and_gradients_vs_reference_golden(golden, dtype, tol, gtol):
    d = golden("wide_512x3_32")
    H, W, _ = d["img"].shape
    p = so.unflatten(d["init"], 512, 3)
    eng = _engine(H, W, 512, 3, dtype, p, torch.tensor(d["img"]))
    pred, sse = eng.forward()
    assert np.abs(pred.cpu().numpy() - d["pred"]).max() <= tol
    assert abs(sse / (3 * H * W) - float(d["loss"])) <= 50 * tol * float(d["loss"])
    eng.forward_backward()
    g, ref = eng.get_grads().cpu().numpy(), d["grads"]
    assert np.linalg.norm(g - ref) <= gtol * np.linalg.norm(ref)
    off = 0
    for fin, fout in so.layer_dims(512, 3):
        for n in (fin * fout, fout):
            a, b = g[off:off + n], ref[off:off + n]
            off += n
            assert np.linalg.norm(a - b) <= gtol * np.linalg.norm(b), (fin, fout, n)
    if dtype == "f16":   # 10 Adam steps against the reference's loss curve (before trajectories decorrelate)
        eng.set_params(torch.tensor(d["init"]).cuda())
        losses = np.array(eng.step([3e-4] * 10, want_loss=True))
        assert np.abs(losses / d["losses"] - 1).max() <= 2e-2


@pytest.mark.parametrize("H,W,hidden,depth,chunk", [(40, 52, 512, 5, 0), (9, 33, 1024, 3, 0), (48, 48, 1024, 4, 1024),
                                                     (1, 1, 512, 3, 0)])
def test_wide_vs_oracle(H, W, hidden, depth, chunk):
    """gradients of every layer against the fp32 oracle (itself pinned to the reference at 512x3), incl. ragged
    grids and a multi-chunk pass; tolerances as for the narrow widths (fp16 operands)."""
    p = so.siren_init(hidden, depth, seed=3)
    img = so.synthetic_image(H, W, seed=5)
    eng = _engine(H, W, hidden, depth, "f16", p, img, chunk_pixels=chunk)
    grid = so.get_grid(H, W)
    pred, sse = eng.forward()
    assert np.abs(pred.cpu().numpy() - so.forward(p, grid).numpy()).max() <= 5e-4
    loss, sse_ref, grads = so.loss_and_grads(p, grid, img)
    assert abs(sse - sse_ref) <= 2e-3 * sse_ref
    eng.forward_backward()
    g, ref = eng.get_grads().cpu().numpy(), so.flatten(grads)
    off = 0
    for fin, fout in so.layer_dims(hidden, depth):
        for n in (fin * fout, fout):
            a, b = g[off:off + n], ref[off:off + n]
            off += n
            assert np.linalg.norm(a - b) <= 6e-3 * np.linalg.norm(b) + 1e-12, (fin, fout, n)


def test_wide_host_mirror_padded_width():
    """Siren(hidden_size=300) runs zero-padded on the 512-wide kernels; one train_epoch equals the oracle's."""
    from implicit_image.models import Siren
    from implicit_image.utils import train_helper as th
    from implicit_image.data import get_grid
    torch.manual_seed(0)
    m = Siren(depth=3, hidden_size=300, first_omega_0=50.0, hidden_omega_0=30.0).cuda()
    p = [q.detach().cpu().clone() for q in m._param_list()]
    H, W = 24, 40
    img = so.synthetic_image(H, W, seed=2)
    grid = get_grid(H, W).cuda()
    optim, sched = th.get_optimizer_lr_scheduler(m, dict(name="adam", lr=3e-4))
    loss = th.train_epoch(m, optim, grid, img.cuda(), lr_scheduler=sched)
    opt = so.Adam(p)
    ref = so.train_epoch(p, opt, so.get_grid(H, W), img, 0)
    assert abs(loss - ref) <= 2e-3 * ref
    m.download_params()
    for a, b in zip(m._param_list(), p):
        assert (a.detach().cpu() - b).abs().max() <= 6.2e-4   # one Adam step moves <= lr; sign flips on ~0 gradients


def test_graph_replay_equals_eager_steps():
    """sf_set_graph_replay(1): sf_step replays a captured hipGraph per step; one-step calls run eagerly, and so
    do multi-step calls on a default handle (per-step losses through a device table, one read-back).
    Same kernels in the same order => parameters, Adam state and losses are bit-identical."""
    H, W, hidden, depth = 40, 56, 64, 4
    p = so.siren_init(hidden, depth, seed=1)
    img = so.synthetic_image(H, W, seed=4)
    lrs = [3e-4 * (0.5 ** (t // 5)) for t in range(13)]
    a = _engine(H, W, hidden, depth, "f16", p, img)
    b = _engine(H, W, hidden, depth, "f16", p, img)
    c = _engine(H, W, hidden, depth, "f16", p, img)
    a.set_graph_replay(True)
    assert c.step(lrs + lrs[:6], want_loss=True) == [b.step([lr], want_loss=True)[0] for lr in lrs + lrs[:6]]
    assert torch.equal(c.get_params(), b.get_params())
    b.set_params(torch.tensor(so.flatten(p)).cuda())
    b.set_adam_state(torch.zeros(b.num_params, device="cuda"), torch.zeros(b.num_params, device="cuda"), 0)
    la = a.step(lrs, want_loss=True)                       # replay
    la += a.step(lrs[:6], want_loss=True)                  # cached graph, fewer steps
    lb = [b.step([lr], want_loss=True)[0] for lr in lrs + lrs[:6]]
    assert la == lb
    assert torch.equal(a.get_params(), b.get_params())
    ma, va, sa = a.get_adam_state()
    mb, vb, sb = b.get_adam_state()
    assert sa == sb == 19 and torch.equal(ma, mb) and torch.equal(va, vb)
    mask = (torch.rand(a.num_params, device="cuda") > 0.5).float()
    a.set_masks(mask); b.set_masks(mask)                    # mask toggles -> the graph is re-captured
    a.step(lrs[:4]); [b.step([lr]) for lr in lrs[:4]]
    assert torch.equal(a.get_params(), b.get_params())
    pa, _ = a.forward()
    pb, _ = b.forward()
    assert torch.equal(pa, pb)


@pytest.mark.parametrize("with_mask", [False, True])
def test_train_steps_bulk_equals_step_by_step(with_mask):
    """train_steps(n) (one sf_step call between host events, used by fit_one) against n train_epoch() calls:
    losses, weights, masks, prune-rate schedule and learning-rate schedule must be bit-identical."""
    from implicit_image.data import get_grid
    from implicit_image.models import registry
    from implicit_image.utils.train_helper import get_optimizer_lr_scheduler, setup_mask, train_epoch, train_steps

    class Cfg(dict):
        __getattr__ = dict.get
    H, W = 48, 40
    img, grid = so.synthetic_image(H, W, seed=9).cuda(), get_grid(H, W).cuda()
    runs = []
    for bulk in (False, True):
        torch.manual_seed(0)
        model = registry["siren"](depth=4, hidden_size=64, first_omega_0=50, hidden_omega_0=30).to("cuda")
        optim, sched = get_optimizer_lr_scheduler(model, Cfg(name="adam", lr=3e-4))
        sched.step_size = 7                                   # exercise the StepLR boundary inside a bulk call
        mask = None
        if with_mask:
            mcfg = Cfg(name="RigL", density=0.5, sparse_init="erdos-renyi-kernel", dense_gradients=True,
                       growth_mode="absolute-gradient", prune_mode="magnitude", redistribution_mode="none",
                       dense=False, prune_rate=0.1, decay_schedule="cosine", end_when=30, interval=10)
            mask = setup_mask(model, optim, mcfg)
        losses = []
        for first in range(0, 30, 10):
            if bulk:
                losses += train_steps(model, optim, grid, img, 10, lr_scheduler=sched, mask=mask)
            else:
                losses += [train_epoch(model, optim, grid, img, lr_scheduler=sched, mask=mask) for _ in range(10)]
            if mask:
                mask.update_connections()
        runs.append((losses, [p.detach().clone() for p in model._param_list()], optim.param_groups[0]["lr"],
                     mask.prune_rate if mask else None, mask.mask_step if mask else None,
                     [mask.mask_dict[k].clone() for k in sorted(mask.mask_dict)] if mask else []))
    (la, pa, lra, ra, sa, ma), (lb, pb, lrb, rb, sb, mb) = runs
    assert [float(np.float32(x)) for x in la] == [float(np.float32(x)) for x in lb]
    assert lra == lrb and ra == rb and sa == sb
    for x, y in zip(pa + ma, pb + mb):
        assert torch.equal(x, y)


@pytest.mark.parametrize("hidden,depth,H,W", [(64, 3, 16, 20), (512, 3, 12, 40)])
def test_sine_output_layer(golden, hidden, depth, H, W):
    """outermost_linear=False (siren.py:110-117): the output layer is a sine too.  64x3 against the reference's
    golden vector (incl. its 10-step loss curve), 512x3 (wide kernels) against the oracle."""
    if hidden == 64:
        d = golden("sine_out_64x3_16")
        p, img = so.unflatten(d["init"], hidden, depth), torch.tensor(d["img"])
    else:
        p, img = so.siren_init(hidden, depth, seed=2), so.synthetic_image(H, W, seed=6)
    grid = so.get_grid(H, W)
    eng = _engine(H, W, hidden, depth, "f16", p, img, outermost_linear=False)
    pred, sse = eng.forward()
    assert np.abs(pred.cpu().numpy() - so.forward(p, grid, outermost_linear=False).numpy()).max() <= 6e-4
    loss, sse_ref, grads = so.loss_and_grads(p, grid, img, outermost_linear=False)
    assert abs(sse - sse_ref) <= 3e-3 * sse_ref
    eng.forward_backward()
    g, ref = eng.get_grads().cpu().numpy(), so.flatten(grads)
    assert np.linalg.norm(g - ref) <= 6e-3 * np.linalg.norm(ref)
    if hidden == 64:
        assert np.linalg.norm(g - d["grads"]) <= 6e-3 * np.linalg.norm(d["grads"])
        losses = np.array(eng.step([3e-4] * 10, want_loss=True))
        assert np.abs(losses / d["losses"] - 1).max() <= 2e-2


def test_wide_psnr_parity_after_equal_steps():
    """north-star criterion on the wide kernels: SIREN 512x4 on 96x96, 60 full-batch Adam steps from the same
    seed-0 init: PSNR within 0.05 dB of the fp32 oracle (a 3e-4 relative perturbation of the initial weights moves
    the oracle's own PSNR by 0.001 dB at this point of the trajectory, 0.02 dB at 150 steps)."""
    H = W = 96
    hidden, depth, steps = 512, 4, 60
    img, grid = so.synthetic_image(H, W, seed=8), so.get_grid(H, W)
    p = so.siren_init(hidden, depth, seed=0)
    eng = _engine(H, W, hidden, depth, "f16", p, img)
    got = np.array(eng.step([so.step_lr(3e-4, t) for t in range(steps)], want_loss=True))
    opt = so.Adam(p)
    ref = np.array([so.train_epoch(p, opt, grid, img, t) for t in range(steps)])
    assert np.abs(got / ref - 1).max() <= 1e-2
    _, sse = eng.forward(want_pred=False)
    psnr = 10 * math.log10(3 * H * W / sse)
    _, _, psnr_ref, _ = so.eval_epoch(p, grid, img)
    assert abs(psnr - psnr_ref) <= 0.05, (psnr, psnr_ref)


def test_wide_pixel_split_handles_compose():
    """row-sharded handles on the wide kernels (config 5 shape in miniature): shard SSE and gradients add up."""
    H, W, hidden, depth = 48, 40, 512, 4
    p = so.siren_init(hidden, depth, seed=0)
    img = so.synthetic_image(H, W, seed=4)
    full = _engine(H, W, hidden, depth, "f16", p, img)
    sse = full.forward_backward()
    g = full.get_grads()
    tot, gs = 0.0, torch.zeros_like(g)
    for r0, r1 in ((0, 13), (13, 48)):
        part = _engine(H, W, hidden, depth, "f16", p, img, row_begin=r0, row_end=r1)
        tot += part.forward_backward()
        gs += part.get_grads()
    assert abs(tot - sse) <= 1e-6 * sse
    assert (gs - g).norm().item() <= 1e-5 * g.norm().item()


def test_set_coords_rejects_non_linspace_vectors():
    """the backward of layers 0/1 re-derives pixel coordinates as i/(n-1): vectors that are not get_grid()'s
    linspace(0, 1, n) must be refused instead of silently training on inconsistent coordinates."""
    from implicit_image._engine import SirenEngine
    eng = SirenEngine(8, 12, 64, 3)
    eng.set_coords(torch.linspace(0, 1, 8).cuda(), torch.linspace(0, 1, 12).cuda())
    with pytest.raises(RuntimeError, match="linspace"):
        eng.set_coords((torch.linspace(0, 1, 8) ** 2).cuda(), torch.linspace(0, 1, 12).cuda())
    one = SirenEngine(1, 1, 64, 3)                       # a 1x1 grid: linspace(0, 1, 1) == [0]
    one.set_coords(torch.zeros(1).cuda(), torch.zeros(1).cuda())


def test_metric_size_shard_composition_and_determinism():
    """BASELINE metric configuration at its FULL size (SIREN 256x8, 4096x4096x3, four 4 Mi-pixel chunks):
    size-independent properties instead of an oracle run - two row-sharded handles (the pixel-split mode) add up
    to the full-image SSE and gradient, and a second pass reproduces the first bit for bit."""
    H = W = 4096
    p = so.siren_init(256, 8, seed=0)
    ys = torch.linspace(0, 1, H, device="cuda")[:, None, None]
    xs = torch.linspace(0, 1, W, device="cuda")[None, :, None]
    k = torch.tensor([1.0, 2.0, 3.0], device="cuda")
    img = (0.5 + 0.25 * torch.sin(12 * xs * k) + 0.25 * torch.cos(9 * ys * k)).contiguous()
    full = _engine(H, W, 256, 8, "f16", p)
    full.set_target(img)
    sse = full.forward_backward()
    g = full.get_grads().clone()
    assert math.isfinite(sse) and torch.isfinite(g).all()
    assert full.forward_backward() == sse and torch.equal(full.get_grads(), g)
    full.close()
    tot, gs = 0.0, torch.zeros_like(g)
    for r0, r1 in ((0, 2048), (2048, 4096)):
        part = _engine(H, W, 256, 8, "f16", p, row_begin=r0, row_end=r1)
        part.set_target(img[r0:r1].contiguous())
        tot += part.forward_backward()
        gs += part.get_grads()
        part.close()
    assert abs(tot - sse) <= 1e-6 * sse
    assert (gs - g).norm().item() <= 1e-5 * g.norm().item()


def test_config5_shape_shard_composition_and_determinism():
    """BASELINE config 5 network (SIREN 1024x12, wide kernels) on a 1024x1024 slice of its grid: the row-sharded
    handles of the pixel-split mode add up to the full SSE / gradient; a second pass is bit-identical."""
    H = W = 1024
    p = so.siren_init(1024, 12, seed=0)
    ys = torch.linspace(0, 1, H, device="cuda")[:, None, None]
    xs = torch.linspace(0, 1, W, device="cuda")[None, :, None]
    k = torch.tensor([1.0, 2.0, 3.0], device="cuda")
    img = (0.5 + 0.25 * torch.sin(12 * xs * k) + 0.25 * torch.cos(9 * ys * k)).contiguous()
    full = _engine(H, W, 1024, 12, "f16", p, chunk_pixels=1 << 19)      # two chunks
    full.set_target(img)
    sse = full.forward_backward()
    g = full.get_grads().clone()
    assert math.isfinite(sse) and torch.isfinite(g).all()
    assert full.forward_backward() == sse and torch.equal(full.get_grads(), g)
    full.close()
    tot, gs = 0.0, torch.zeros_like(g)
    for r0, r1 in ((0, 512), (512, 1024)):
        part = _engine(H, W, 1024, 12, "f16", p, row_begin=r0, row_end=r1, chunk_pixels=1 << 19)
        part.set_target(img[r0:r1].contiguous())
        tot += part.forward_backward()
        gs += part.get_grads()
        part.close()
    assert abs(tot - sse) <= 1e-6 * sse
    assert (gs - g).norm().item() <= 1e-5 * g.norm().item()


# ---------------------------------------------------------------------------------------------------------
# round 2: the metric model pinned at a plateau, the remaining BASELINE shapes, the index paths on the device
# ---------------------------------------------------------------------------------------------------------
def _fit_plateau(dtype, fmt, d):
    S, steps, lr_step = int(d["size"]), int(d["steps"]), int(d["lr_step"])
    img = so.synthetic_image(S, S)
    p = so.siren_init(256, 8, seed=0)
    assert np.array_equal(so.flatten(p)[:64], d["init_head"])          # the fixture's seed-0 init
    eng = _engine(S, S, 256, 8, dtype, p, img, scratch_format=fmt)
    losses = np.array(eng.step([3e-4 * 0.5 ** (t // lr_step) for t in range(steps)], want_loss=True))
    _, sse = eng.forward(want_pred=False)
    return 10 * math.log10(3 * S * S / sse), losses


@pytest.mark.parametrize("fmt", FORMATS)
def test_psnr_parity_at_the_metric_model_plateau(golden, fmt):
    """north-star criterion at SIREN 256x8 (BASELINE metric model): 200 full-batch steps, Adam lr 3e-4 halved every
    40 steps, on the 256x256 formula image, against the REAL reference (tests/golden/plateau_256x8_256.npz; its own
    8-vs-2-thread PSNR spread on this run is 0.0000 dB, stored in the fixture).  |dPSNR| <= 0.05 dB for every
    fp16 scratch format (measured on MI355X: +0.016 / +0.012 / +0.002 dB for formats 16 / 12 / 8)."""
    d = golden("plateau_256x8_256")
    assert float(d["psnr_spread"]) <= 0.02
    psnr, losses = _fit_plateau("f16", fmt, d)
    assert abs(psnr - float(d["psnr"])) <= 0.05, (psnr, float(d["psnr"]))
    # the first steps of this fit amplify perturbations ~5x per step (1e-3 at step 3, 5e-3 .. 9e-3 at step 4)
    assert np.max(np.abs(losses[:4] - d["losses"][:4]) / d["losses"][:4]) <= 3e-3
    assert abs(losses[-1] / d["losses"][-1] - 1) <= 2e-2


@pytest.mark.parametrize("fmt", FORMATS)
def test_psnr_parity_at_the_metric_model_plateau_512(golden, fmt):
    """The same criterion on four times the pixels (512x512, tests/golden/plateau_256x8_512.npz: the REAL reference, 8- and
    4-thread runs agree to 0.0000 dB): |dPSNR| <= 0.05 dB.  Measured on MI355X: +0.003 / -0.002 / -0.0002 dB for formats
    16 / 12 / 8 against +0.016 / +0.012 / +0.002 dB on 256x256 - the deviation falls with the pixel count, which is what the
    auto rule (fp8 deltas from 2^20 pixels) relies on."""
    d = golden("plateau_256x8_512")
    assert float(d["psnr_spread"]) <= 0.02
    psnr, losses = _fit_plateau("f16", fmt, d)
    assert abs(psnr - float(d["psnr"])) <= 0.05, (psnr, float(d["psnr"]))
    assert abs(psnr - float(d["psnr"])) <= 0.02, (psnr, float(d["psnr"]))     # measured <= 0.003: keep the trend visible
    assert np.max(np.abs(losses[:4] - d["losses"][:4]) / d["losses"][:4]) <= 3e-3
    assert abs(losses[-1] / d["losses"][-1] - 1) <= 2e-2


@pytest.mark.parametrize("fmt", (16, 12, 8, 0))
def test_psnr_parity_at_the_metric_model_plateau_1024(golden, fmt):
    """The criterion at 2^20 pixels, where scratch_format 0 resolves to fp8 deltas: SIREN 256x8 on the 1024x1024 formula image,
    200 annealed steps, against the REAL reference (tests/golden/plateau_256x8_1024.npz, one 8-thread run of 70 minutes:
    PSNR 30.8025 dB; its thread-count spread is 0.0000 dB on the two smaller fixtures).  Measured on MI355X:
    -0.0025 / -0.0035 / -0.0026 dB for formats 16 / 12 / 8 (auto = 8)."""
    d = golden("plateau_256x8_1024")
    S, steps, lr_step = int(d["size"]), int(d["steps"]), int(d["lr_step"])
    img = so.synthetic_image(S, S)
    p = so.siren_init(256, 8, seed=0)
    assert np.array_equal(so.flatten(p)[:64], d["init_head"])
    eng = _engine(S, S, 256, 8, "f16", p, img, scratch_format=fmt)
    assert eng.scratch_format == (fmt or 8)
    losses = np.array(eng.step([3e-4 * 0.5 ** (t // lr_step) for t in range(steps)], want_loss=True))
    _, sse = eng.forward(want_pred=False)
    psnr = 10 * math.log10(3 * S * S / sse)
    assert abs(psnr - float(d["psnr"])) <= 0.02, (psnr, float(d["psnr"]))      # criterion 0.05; measured <= 0.004
    assert np.max(np.abs(losses[:4] - d["losses"][:4]) / d["losses"][:4]) <= 3e-3
    assert abs(losses[-1] / d["losses"][-1] - 1) <= 2e-2


def test_bf16_operands_miss_the_plateau_criterion(golden):
    """Why compute_dtype defaults to fp16 although BASELINE.json says bf16: same run, bf16 operands.  Measured
    +0.081 dB (8-bit significands perturb every step's gradient by ~0.4 %); the assertion records that it is outside
    the 0.05 dB criterion the fp16 formats meet, and that it is not wildly off (a kernel bug would be)."""
    d = golden("plateau_256x8_256")
    psnr, _ = _fit_plateau("bf16", 16, d)
    dev = abs(psnr - float(d["psnr"]))
    assert 0.05 < dev <= 0.3, dev


def _strided_grad_check(g, d, hidden, depth, tol):
    """fixture gradients are stored every grad_stride-th element plus per-tensor L2 norms"""
    stride = int(d["grad_stride"])
    assert _rel(g[::stride], d["grads"]) <= tol
    off = 0
    norms = []
    for fin, fout in so.layer_dims(hidden, depth):
        for n in (fin * fout, fout):
            norms.append(float(np.linalg.norm(g[off:off + n])))
            off += n
    assert np.max(np.abs(np.array(norms) / d["grad_norms"] - 1)) <= 2 * tol


@pytest.mark.parametrize("hidden,depth,fmt", [(256, 6, 16), (256, 6, 12), (256, 6, 8), (512, 6, 0), (512, 8, 0)])
def test_config3_shapes_vs_reference_golden(golden, hidden, depth, fmt):
    """BASELINE config 3 networks ({256,512} x {6,8}; 256x8 is covered above) on a ragged 24x40 image: prediction,
    loss, first-step gradient and the first 10 Adam steps against the real reference."""
    from oracle import engine_model as em
    d = golden(f"shapes_{hidden}x{depth}")
    H, W, _ = d["img"].shape
    p = so.siren_init(hidden, depth, seed=0)
    assert np.array_equal(so.flatten(p)[:64], d["init_head"])
    img = torch.tensor(d["img"])
    eng = _engine(H, W, hidden, depth, "f16", p, img, scratch_format=fmt)
    pred, sse = eng.forward()
    assert np.abs(pred.cpu().numpy() - d["pred"]).max() <= (3e-4 if hidden <= 256 else 6e-4)
    assert abs(sse / (3 * H * W) - float(d["loss"])) <= 2e-3 * float(d["loss"])
    eng.forward_backward()
    g = eng.get_grads().cpu().numpy()
    if fmt in (12, 8):   # byte formats: the numerics model's own noise against the fp32 reference sets the bound
        model = so.flatten(em.loss_and_grads(p, so.get_grid(H, W), img, scratch=fmt)[2])
        tol = 1.5 * _rel(model[::int(d["grad_stride"])], d["grads"]) + 2e-3
    else:
        tol = 5e-3
    _strided_grad_check(g, d, hidden, depth, tol)
    losses = np.array(eng.step([3e-4] * 10, want_loss=True))
    # the first steps of these over-parameterised fits amplify perturbations (see test_short_run_256x8_vs_reference)
    assert np.max(np.abs(losses[:3] - d["losses"][:3]) / d["losses"][:3]) <= (3e-3 if fmt != 8 else 1e-2)
    assert losses[-1] < losses[0]


def test_rigl_at_config4_shape_vs_reference_golden(golden):
    """BASELINE config 4 at its real shape: SIREN 256x8, RigL density 0.1 (ERK), 160 steps on 48x48 (Adam lr 3e-4
    halved every 40 steps: the PSNR compared is a settled value), topology updates at i = 0, 20, 40, 60, 80
    (compress.py:141-143).  mask0 bit-exact; the per-layer non-zero budget after every update, the density trace and
    the prune-rate positions exact; PSNR within 0.05 dB of the reference, whose OWN 8-vs-2-thread runs differ by
    0.0004 dB while already disagreeing on the masks (fixture: masks_equal_2threads = False)."""
    from implicit_image.data import get_grid
    from implicit_image.models import registry
    from implicit_image.utils.train_helper import eval_epoch, get_optimizer_lr_scheduler, setup_mask, train_epoch
    d = golden("rigl_256x8_48")

    class Cfg(dict):
        __getattr__ = dict.get
    H = W = 48
    torch.manual_seed(0)
    model = registry["siren"](depth=8, hidden_size=256, first_omega_0=50, hidden_omega_0=30).to("cuda")
    optim, _ = get_optimizer_lr_scheduler(model, Cfg(name="adam", lr=3e-4))
    sched = torch.optim.lr_scheduler.StepLR(optim, int(d["lr_step"]), gamma=0.5)
    mcfg = Cfg(name="RigL", density=0.1, sparse_init="erdos-renyi-kernel", dense_gradients=True,
               growth_mode="absolute-gradient", prune_mode="magnitude", redistribution_mode="none", dense=False,
               prune_rate=0.1, decay_schedule="cosine", end_when=90, interval=20)
    mask = setup_mask(model, optim, mcfg)
    assert model.cfg["scratch_format"] == 16          # topology decisions read the dense gradient: unorm16 phases
    names = [n for n, _ in model.named_parameters() if n in mask.mask_dict]

    def bits():
        return np.packbits(np.concatenate([mask.mask_dict[n].cpu().numpy().ravel().astype(np.uint8) for n in names]))
    assert np.array_equal(bits(), d["mask0"])
    img, grid = torch.tensor(d["img"]).cuda(), get_grid(H, W).cuda()
    losses, dens, u = [], [], 0
    for i in range(int(d["steps"])):
        losses.append(train_epoch(model, optim, grid, img, lr_scheduler=sched, mask=mask))
        if i <= mcfg.end_when and i % mcfg.interval == 0:
            assert mask.prune_rate == pytest.approx(float(d[f"upd{u}_rate"]), rel=0, abs=1e-12)
            mask.update_connections()
            assert [int(mask.mask_dict[n].sum().item()) for n in names] == d[f"upd{u}_nnz"].tolist()   # budget per layer
            u += 1
        dens.append(mask.stats.total_density)
    assert u == int(d["n_updates"]) and mask.mask_step == int(d["mask_step"])
    assert np.allclose(dens, d["density"], rtol=0, atol=1e-12)
    assert np.max(np.abs(np.array(losses[:5]) - d["losses"][:5]) / d["losses"][:5]) <= 3e-3
    _, _, psnr, _ = eval_epoch(model, grid, img)
    assert abs(psnr - float(d["psnr"])) <= 0.05, (psnr, float(d["psnr"]), float(d["psnr_spread"]))
    for n, w in model.named_parameters():
        if n in mask.mask_dict:
            assert torch.all(w.data[mask.mask_dict[n] == 0] == 0)


@pytest.mark.parametrize("hidden,depth,density", [(64, 4, 0.5), (256, 8, 0.1)])
def test_truncate_weights_bit_exact_on_the_device(golden, hidden, depth, density):
    """The reference's own (w, grad, mask, rate) -> (mask', w') pair replayed with every tensor ON THE GPU and bound to
    the engine (torch.sort / comparisons run as device kernels there): magnitude prune + absolute-gradient growth must
    reproduce the reference's masks and weights bit for bit (prune.py:24-51, grow.py:58-97)."""
    from implicit_image.data import get_grid
    from implicit_image.models import registry
    from implicit_image.utils.train_helper import get_optimizer_lr_scheduler, setup_mask
    d = golden(f"truncate_{hidden}x{depth}")

    class Cfg(dict):
        __getattr__ = dict.get
    torch.manual_seed(0)
    m = registry["siren"](depth=depth, hidden_size=hidden, first_omega_0=50, hidden_omega_0=30).to("cuda")
    opt, _ = get_optimizer_lr_scheduler(m, Cfg(name="adam", lr=3e-4))
    mask = setup_mask(m, opt, Cfg(name="RigL", density=density, sparse_init="erdos-renyi-kernel", dense_gradients=True,
                                  growth_mode="absolute-gradient", prune_mode="magnitude", redistribution_mode="none",
                                  dense=False, prune_rate=0.1, decay_schedule="cosine", end_when=1500, interval=20))
    m.engine(get_grid(16, 16).cuda())            # bind: weight.data / weight.grad become views of engine memory
    off, moff, mbits = 0, 0, np.unpackbits(d["mask_in"])
    for n, p in m.named_parameters():
        k = p.numel()
        assert p.data.is_cuda and p.grad is not None and p.grad.is_cuda
        p.data.copy_(torch.tensor(d["w_in"][off:off + k]).view(p.shape))
        p.grad.copy_(torch.tensor(d["g_in"][off:off + k]).view(p.shape))
        off += k
        if n in mask.mask_dict:
            mask.mask_dict[n] = torch.tensor(mbits[moff:moff + k].astype(np.float32)).view(p.shape).cuda()
            moff += k
    mask.mask_step = int(d["mask_step_in"])
    for s in range(mask.mask_step):
        mask.prune_rate_decay.step(s)
    mask.update_connections()
    names = [n for n, _ in m.named_parameters() if n in mask.mask_dict]
    got = np.packbits(np.concatenate([mask.mask_dict[n].cpu().numpy().ravel().astype(np.uint8) for n in names]))
    assert np.array_equal(got, d["mask_out"])
    flat = np.concatenate([p.detach().cpu().numpy().ravel() for p in m.parameters()]).astype(np.float32)
    assert np.array_equal(flat, d["w_out"])


def test_config5_one_rank_shard_of_the_8192_grid():
    """BASELINE config 5 as ONE of its 8 ranks sees it: SIREN 1024x12 on rows [0, 1024) of the 8192 x 8192 grid
    (W = 8192 exercises the 2^40 / W row-column decode and the 64-bit piece offsets; 8 Mi pixels in eight 1 Mi-pixel
    chunks).  Size-independent properties: a second pass is bit-identical; the two 512-row halves add up to the
    shard's SSE and gradient."""
    H = W = 8192
    R = 1024
    p = so.siren_init(1024, 12, seed=0)
    ys = torch.linspace(0, 1, H, device="cuda")[:R, None, None]
    xs = torch.linspace(0, 1, W, device="cuda")[None, :, None]
    k = torch.tensor([1.0, 2.0, 3.0], device="cuda")
    img = (0.5 + 0.25 * torch.sin(12 * xs * k) + 0.25 * torch.cos(9 * ys * k)).contiguous()
    full = _engine(H, W, 1024, 12, "f16", p, row_begin=0, row_end=R)
    full.set_target(img)
    sse = full.forward_backward()
    g = full.get_grads().clone()
    assert math.isfinite(sse) and torch.isfinite(g).all() and g.abs().max().item() > 0
    assert full.forward_backward() == sse and torch.equal(full.get_grads(), g)
    full.close()
    tot, gs = 0.0, torch.zeros_like(g)
    for r0, r1 in ((0, 512), (512, 1024)):
        part = _engine(H, W, 1024, 12, "f16", p, row_begin=r0, row_end=r1)
        part.set_target(img[r0:r1].contiguous())
        tot += part.forward_backward()
        gs += part.get_grads()
        part.close()
    assert abs(tot - sse) <= 1e-6 * sse
    assert (gs - g).norm().item() <= 1e-5 * g.norm().item()
