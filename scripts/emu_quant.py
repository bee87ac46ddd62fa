#!/usr/bin/env python3
"""CPU emulation of candidate scratch-tensor encodings of the engine's backward pass (round 2).

Question answered before any kernel is written: does the fit still land within 0.05 dB of the fp32
reference when the tensors the backward re-reads from HBM are narrower?

  phases  : unorm16 (round-1 engine) | 12-bit | 8-bit          (k_fwd spill, decoded by k_bwd for sin AND cos)
  deltas  : fp16 (round-1 engine) | fp8 e4m3 with one power-of-two scale per tensor ("global")
            | fp8 e4m3 with one power-of-two scale per (32 pixels x 32 neurons) tile ("tile")

Everything else follows oracle/engine_model.py (fp16 GEMM operands, fp32 accumulation, layer 0 in fp32).
The fp32 baseline is the same explicit forward/backward with no rounding anywhere (== the oracle).

    python scripts/emu_quant.py --size 128 --steps 2000 --variants ref,ref_alt,f16,p8,d8g,d8t,p8d8t --procs 4

Prints one JSON line per variant: final PSNR of an fp32 re-evaluation of the trained weights, the PSNR curve at
every 250 steps, and the difference to `ref`.  Results of the runs that decided the round-2 encoding are kept in
profiles/r02_emu_quant.jsonl.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import siren_oracle as so  # noqa: E402

TWO_PI = 2.0 * math.pi


def rt16(x):
    return x.to(torch.float16).float()


def q_phase(t, bits):
    """phase in revolutions -> value the backward decodes (revolutions in [0,1))."""
    fr = t - torch.floor(t)
    if bits == 16:   # v_cvt_pknorm_u16: round(x*65535), decoded as u/65536
        return torch.round(fr * 65535.0).clamp_(0, 65535) * (1.0 / 65536.0)
    n = float(1 << bits)   # round(t * 2^bits) mod 2^bits, decoded as u / 2^bits
    return torch.remainder(torch.round(t * n), n) / n


def q_delta(d, mode, gpre):
    """delta tensor [N, W] (true scale) -> what the consumer kernels see."""
    if mode == "f32":
        return d
    if mode == "f16":
        return rt16(d * gpre) / gpre
    x = d * gpre
    if mode == "e4m3_global":
        s = torch.tensor(1.0)
    elif mode == "e4m3_tile":     # one power-of-two scale per 32-pixel x 32-neuron tile: amax -> [224, 448]
        n, w = x.shape
        npad = (n + 31) // 32 * 32
        xp = torch.zeros(npad, w)
        xp[:n] = x
        am = xp.reshape(npad // 32, 32, w // 32, 32).abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-30)
        s = torch.exp2(torch.ceil(torch.log2(am / 448.0))).expand(npad // 32, 32, w // 32, 32).reshape(npad, w)[:n]
    elif mode == "e5m2_global":
        return (x.clamp(-57344, 57344).to(torch.float8_e5m2).float()) / gpre
    else:
        raise ValueError(mode)
    y = (x / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * s
    return y / gpre


def loss_and_grads(params, x, img, n_total, *, fwd16, pbits, dmode, gpre, om0=50.0, om=30.0):
    depth = len(params) // 2
    rt = rt16 if fwd16 else (lambda v: v)
    ws = 256.0 if fwd16 else 1.0
    W0, b0 = params[0], params[1]
    z = torch.addcmul(torch.addcmul(b0, x[:, 0:1], W0[:, 0]), x[:, 1:2], W0[:, 1])
    t = z * (om0 / TWO_PI)
    ph = [t - torch.floor(t)]     # layer 0: recomputed in the backward, not quantised
    a = torch.sin(TWO_PI * (t - torch.floor(t)))
    hs = om / TWO_PI
    for l in range(1, depth - 1):
        t = rt(a) @ rt(params[2 * l] * hs).t() + params[2 * l + 1] * hs
        ph.append(t - torch.floor(t) if pbits == 0 else q_phase(t, pbits))
        a = torch.sin(TWO_PI * (t - torch.floor(t)))
    L = depth - 1
    out = (rt(a) @ rt(params[2 * L] * ws).t() + params[2 * L + 1] * ws) * (1.0 / ws)
    pred = out * 0.5 + 0.5
    resid = pred - img
    sse = float((resid.double() ** 2).sum())
    g = resid * (1.0 / (3.0 * n_total))           # d mse / d out = 2 r / 3N, times the 1/2 of siren.py:131
    delta = q_delta(g, "f16" if fwd16 else "f32", gpre)   # dL/dout stays fp16 (64 B per pixel)
    grads = [None] * (2 * depth)
    for l in range(L, 0, -1):
        p = ph[l - 1]
        act = rt(torch.sin(TWO_PI * p))
        grads[2 * l] = delta.t() @ act
        grads[2 * l + 1] = delta.sum(0)
        omm = om0 if l - 1 == 0 else om
        G = delta @ rt(params[2 * l] * omm)
        delta = q_delta(G * torch.cos(TWO_PI * p), dmode, gpre)
    grads[0] = delta.t() @ x
    grads[1] = delta.sum(0)
    return sse / (3.0 * n_total), grads


VARIANTS = {
    #            fwd16  pbits dmode
    "ref":      (False, 0, "f32"),
    "ref_alt":  (False, 0, "f32"),          # same arithmetic, other thread count: the reference's own summation-order noise
    "f16":      (True, 16, "f16"),          # round-1 engine
    "p12":      (True, 12, "f16"),
    "p8":       (True, 8, "f16"),
    "d8g":      (True, 16, "e4m3_global"),
    "d8t":      (True, 16, "e4m3_tile"),
    "d5m2":     (True, 16, "e5m2_global"),
    "p8d8g":    (True, 8, "e4m3_global"),
    "p8d8t":    (True, 8, "e4m3_tile"),
    "p12d8t":   (True, 12, "e4m3_tile"),
}


def psnr_fp32(params, x, img):
    with torch.no_grad():
        a = torch.sin(50.0 * (x @ params[0].t() + params[1]))
        for l in range(1, len(params) // 2 - 1):
            a = torch.sin(30.0 * (a @ params[2 * l].t() + params[2 * l + 1]))
        out = (a @ params[-2].t() + params[-1]) * 0.5 + 0.5
        mse = float(((out - img).double() ** 2).mean())
    return 10.0 * math.log10(1.0 / mse)


def run(name, args):
    fwd16, pbits, dmode = VARIANTS[name]
    torch.set_num_threads(args.threads + 1 if name == "ref_alt" else args.threads)
    H = W = args.size
    params = so.siren_init(args.hidden, args.depth, seed=0)
    img = so.synthetic_image(H, W, seed=args.img_seed, noise=args.noise).reshape(-1, 3)
    x = (so.get_grid(H, W).reshape(-1, 2) - 0.5) * 2
    n = H * W
    gpre = float(2.0 ** (math.ceil(math.log2(3.0 * n)) + 2)) if fwd16 else 1.0
    opt = torch.optim.Adam(params, lr=args.lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, args.lr_step, args.lr_gamma)
    curve = {}
    t0 = time.time()
    for it in range(args.steps):
        with torch.no_grad():
            loss, grads = loss_and_grads(params, x, img, n, fwd16=fwd16, pbits=pbits, dmode=dmode, gpre=gpre)
        for p, g in zip(params, grads):
            p.grad = g.reshape(p.shape).contiguous()
        opt.step()
        sched.step()
        if (it + 1) % args.every == 0 or it + 1 == args.steps:
            curve[it + 1] = round(psnr_fp32(params, x, img), 4)
            print(f"# {name} step {it + 1} loss {loss:.3e} psnr {curve[it + 1]:.3f}  ({time.time() - t0:.0f}s)",
                  file=sys.stderr, flush=True)
    return {"variant": name, "size": args.size, "hidden": args.hidden, "depth": args.depth, "steps": args.steps,
            "lr": args.lr, "lr_step": args.lr_step, "lr_gamma": args.lr_gamma, "psnr": curve[args.steps], "curve": curve}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--every", type=int, default=250)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--lr-step", type=int, default=2000)
    ap.add_argument("--lr-gamma", type=float, default=0.5)
    ap.add_argument("--img-seed", type=int, default=1234)
    ap.add_argument("--noise", type=float, default=0.05, help="amplitude of the uniform noise in the target image")
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--variants", default="ref,ref_alt,f16,p8,d8g,d8t,p8d8t")
    ap.add_argument("--one", default=None, help="internal: run one variant in this process")
    args = ap.parse_args()
    if args.one:
        print(json.dumps(run(args.one, args)), flush=True)
        return
    import subprocess
    names = args.variants.split(",")
    pending, running, results = list(names), [], {}
    base = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]]
    while pending or running:
        while pending and len(running) < args.procs:
            nm = pending.pop(0)
            running.append((nm, subprocess.Popen(base + ["--one", nm], stdout=subprocess.PIPE, text=True)))
        for nm, p in list(running):
            if p.poll() is not None:
                out = p.stdout.read().strip().splitlines()
                results[nm] = json.loads(out[-1])
                running.remove((nm, p))
        time.sleep(1.0)
    ref = results.get("ref", {}).get("psnr")
    for nm in names:
        r = results[nm]
        r["d_vs_ref"] = None if ref is None else round(r["psnr"] - ref, 4)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
