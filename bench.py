#!/usr/bin/env python3
"""Headline benchmark: Mpixel-iters/s of one full SIREN fit step (forward + MSE + backward + Adam)
at hidden 256 x depth 8 on a synthetic 4096x4096x3 grid (BASELINE.json `metric`).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; each rank fits its own image —
   per-image sharding, no data-path collective, weak scaling.)

Prints ONE JSON line on rank 0 with the contract keys plus
  roofline      MFMA roofline of the dominant kernel (algorithmic GEMM FLOPs / HIP-event duration)
  cpu_baseline  the CPU oracle (fp32 restatement of the reference step) timed on the host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "implicit-image-compression_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16/fp16 MFMA peak of MI355X (MI355X_MICROARCH.md chip table)


def kernel_sources():
    """Every source libsiren_fit.so is built from: all csrc/*.hip and csrc/*.h (a new kernel file is covered the day it
    is added - round 2's list had missed the file that held the dominant kernel) + the C ABI header."""
    import glob
    csrc = os.path.join(ROOT, "implicit-image-compression_amd", "csrc")
    return sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))) + [os.path.join(ROOT, "include", "siren_fit.h")]


def kernel_source_hash():
    """sha256 over the kernel sources libsiren_fit.so is built from: a PMC traffic file is only valid for the
    kernels it was measured on (scripts/pmc_traffic_json.py records the same hash)."""
    import hashlib
    h = hashlib.sha256()
    for f in kernel_sources():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def pmc_traffic(kernel, args, npix):
    """HBM bytes per launch of `kernel` from the newest committed PMC passes (profiles/r*_pmc_traffic*.json) taken
    on EXACTLY these kernel sources and this launch geometry (one 4 Mi-pixel chunk of SIREN 256x8); otherwise
    (None, reason) - a stale constant is worse than no number."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True)
    if not files:
        return None, "no profiles/r*_pmc_traffic*.json"
    if not (args.hidden == 256 and args.depth == 8 and args.chunk == 0 and npix >= (1 << 22)):
        return None, "PMC passes cover the default 256x8 / 4 Mi-pixel-chunk geometry only"
    here = kernel_source_hash()
    for f in files:
        try:
            pmc = json.load(open(f))
        except Exception:
            continue
        if pmc.get("kernel_source_sha256") != here:
            continue
        if kernel in pmc.get("kernels", {}) and "hbm_bytes_per_launch" in pmc["kernels"][kernel]:
            return pmc["kernels"][kernel]["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    return None, "kernel sources changed since the committed PMC passes (kernel_source_sha256 mismatch)"


def flops_per_pixel_iter(hidden, depth, out_features=3):
    """F = 6*P_w - 4*hidden, P_w = 2W + (D-2)W^2 + 3W (GEMM multiply-adds only)."""
    pw = 2 * hidden + (depth - 2) * hidden * hidden + out_features * hidden
    return 6 * pw - 4 * hidden


def device_image(h, w, device, seed=1234):
    """SURVEY.md §8(d) synthetic target generated on the device: sinusoids + seeded uniform noise."""
    ys = torch.linspace(0, 1, h, device=device)[:, None, None]
    xs = torch.linspace(0, 1, w, device=device)[None, :, None]
    kx = torch.tensor([1.0, 2.0, 3.0], device=device)
    ky = torch.tensor([3.0, 1.0, 2.0], device=device)
    img = 0.5 + 0.25 * torch.sin(12 * xs * kx) + 0.25 * torch.cos(9 * ys * ky)
    g = torch.Generator(device=device).manual_seed(seed)
    img = img + 0.05 * (torch.rand(h, w, 3, device=device, generator=g) * 2 - 1)
    return img.clamp_(0, 1).contiguous()


def cpu_baseline(hidden, depth, size, warm=1, timed=3):
    """The oracle's train step (fp32 restatement of train_helper.py:132-185) on the host cores.
    Bounded sample: `size` x `size` grid (the 4096^2 grid needs ~325 GB of fp32 activations on this
    path); Mpixel-iters/s is size-normalised."""
    from oracle import siren_oracle as so
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)          # the GPU box grants ~16 host cores per GPU; more threads only thrash
    torch.set_num_threads(cores)
    p = so.siren_init(hidden, depth, seed=0)
    img = so.synthetic_image(size, size)
    grid = so.get_grid(size, size)
    opt = so.Adam(p)
    for t in range(warm):
        so.train_epoch(p, opt, grid, img, t)
    t0 = time.perf_counter()
    for t in range(timed):
        so.train_epoch(p, opt, grid, img, warm + t)
    dt = (time.perf_counter() - t0) / timed
    return {"value": size * size / dt / 1e6, "unit": "Mpixel-iters/s", "cores": cores, "kind": "port",
            "sample": f"{timed} full-batch steps of SIREN {hidden}x{depth} on a {size}x{size} grid, fp32 torch CPU "
                      f"oracle, {dt * 1e3:.0f} ms/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # >= 20 timed, >= 5 warm-up steps
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--scratch", type=int, default=0, choices=[0, 8, 12, 16],
                    help="sf_config.scratch_format: 0 = auto (fp16, hidden <= 512: 8 = phase bytes + fp8 deltas from 2^20 pixels; below that 12 = phase bytes + 16-bit deltas at hidden <= 256), 16 = round-1 format")
    ap.add_argument("--no-formats", action="store_true", help="skip the by_scratch_format leg (5 warm-up + 10 timed steps per format after the headline region)")
    ap.add_argument("--cpu-size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the multi-process "
                         "path with several ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on a 1-GPU box: create the process group (RCCL communicator, barrier, all-reduce) "
                         "even at WORLD_SIZE 1 - RCCL refuses two ranks on one device, gloo does not")
    ap.add_argument("--mode", default="per-image", choices=["per-image", "pixel-split"],
                    help="N>1: per-image = one independent fit per GPU (weak scaling, no collective); "
                         "pixel-split = ONE image, rows sharded over ranks, gradient all-reduce over RCCL (strong scaling)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from implicit_image._engine import SirenEngine
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    H = W = args.size
    split = args.mode == "pixel-split" and (world > 1 or args.force_dist)
    r0, r1 = (0, H)
    if split:
        from implicit_image.parallel import PixelSplitFit, shard_rows
        r0, r1 = shard_rows(H, world, rank)
    eng = SirenEngine(H, W, args.hidden, args.depth, compute_dtype=args.dtype, device=local_rank,
                      chunk_pixels=args.chunk, row_begin=r0, row_end=r1, scratch_format=args.scratch)
    # per-image sharding: every rank fits its own synthetic image (seed offset by rank), same init
    from implicit_image.models import Siren   # seed-0 SIREN init, random-init weights
    torch.manual_seed(0)
    init = Siren(depth=args.depth, hidden_size=args.hidden, first_omega_0=50.0, hidden_omega_0=30.0)
    eng.set_params(torch.cat([q.detach().reshape(-1) for q in init.parameters()]).to(dev))
    eng.set_coords(torch.linspace(0, 1, H).to(dev), torch.linspace(0, 1, W).to(dev))
    img = device_image(H, W, dev, seed=1234 + (0 if split else rank))[r0:r1].contiguous()
    eng.set_target(img)
    lr = 3e-4
    if split:
        psf = PixelSplitFit(eng, 3 * H * W)

    def run_steps(n):
        if split:
            for _ in range(n):
                psf.step(lr, want_loss=False)
        else:
            eng.step([lr] * n)          # no host sync inside: losses stay on the device

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run_steps(args.warmup)
    barrier()
    eng.profile(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    rep = eng.profile_report()
    eng.profile(False)
    _, sse = eng.forward(want_pred=False)
    if split:
        t = torch.tensor([sse], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t)
        sse = t.item()
    psnr = 10 * torch.log10(torch.tensor(3.0 * H * W / sse)).item()

    if rank == 0:
        F = flops_per_pixel_iter(args.hidden, args.depth)
        value = (1 if split else world) * H * W * args.steps / dt / 1e6
        kern = {k: v for k, v in rep.items() if v["launches"]}
        dom = max((k for k in kern if kern[k]["flops_per_launch"] > 0), key=lambda k: kern[k]["total_ms"])
        d = kern[dom]
        avg_ms = d["total_ms"] / d["launches"]
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be collected inside this process, so the
        # value comes from the committed rocprofv3 --pmc passes - only if they were taken on these kernel sources
        traffic, traffic_source = pmc_traffic(dom, args, eng.npix)
        ach = d["flops_per_launch"] / (avg_ms * 1e-3) / 1e12
        out = {
            "metric": f"Mpixel-iters/s (fwd+bwd+Adam) @ SIREN-{args.hidden}x{args.depth}",
            "value": value, "unit": "Mpixel-iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if split else "weak",
            "vs_baseline": None, "dtype": ("f16 (fp16 MFMA operands fwd+bwd, f32 accumulate, f32 optimiser state; backward scratch in HBM: "
                      + {8: "phase bytes + fp8 e4m3 deltas", 12: "phase bytes + fp16 deltas", 16: "unorm16 phases + fp16 deltas"}[eng.scratch_format] + ")")
            if args.dtype == "f16" else "bf16",
            "data": "synthetic",
            "config": {"workload": f"siren_{args.hidden}x{args.depth}_fit_step_{H}x{W}x3_grid", "image": f"{H}x{W}x3",
                       "hidden": args.hidden, "depth": args.depth, "sharding": (f"pixel-split rows x{world} + RCCL grad all-reduce" if split else f"per-image x{world}"),
                       "chunk_pixels": min(eng.npix, args.chunk or (1 << 22) // max(1, args.hidden // 256)),
                       "scratch_format": eng.scratch_format},   # 8 = phase bytes + fp8 deltas, 12 = + 16-bit deltas, 16 = unorm16 / 16-bit
            "step_mfma_frac": value / world * 1e6 * F / (PEAK_BF16_TFLOPS * 1e12),   # per GPU
            "psnr_after_run": psnr,
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": d["bytes_per_launch"],
                         "avg_launch_ms": avg_ms, "flops_per_launch": d["flops_per_launch"]},
            # the same kernel against the HBM roof (it moves 1.5 KiB per pixel per layer: delta and phase in,
            # delta out): algorithmic bytes per launch / launch time vs the 8 TB/s of MI355X_MICROARCH.md
            "roofline_hbm": {"bound": "hbm", "kernel": dom, "achieved": d["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9,
                             "peak": 8000.0, "unit": "GB/s",
                             "frac": d["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic},
            "kernels": {k: {"ms_per_step": v["total_ms"] / args.steps, "launches_per_step": v["launches"] / args.steps,
                            "tflops": (v["flops_per_launch"] * v["launches"] / (v["total_ms"] * 1e-3) / 1e12)
                            if v["total_ms"] > 0 else 0.0,
                            "algorithmic_GBps": (v["bytes_per_launch"] * v["launches"] / (v["total_ms"] * 1e-3) / 1e9)
                            if v["total_ms"] > 0 else 0.0} for k, v in kern.items()},
        }
        if world == 1 and not args.no_formats and args.hidden <= 256 and args.dtype == "f16":
            # the same workload with every backward scratch format forced, timed by the same clock AFTER the headline
            # region (5 warm-up + 10 timed steps each): the 16-bit-scratch throughput next to the fp8 one
            del eng
            torch.cuda.empty_cache()
            out["by_scratch_format"] = {}
            for fmt in (8, 12, 16):
                e = SirenEngine(H, W, args.hidden, args.depth, compute_dtype=args.dtype, device=local_rank, chunk_pixels=args.chunk, scratch_format=fmt)
                e.set_params(torch.cat([q.detach().reshape(-1) for q in init.parameters()]).to(dev))
                e.set_coords(torch.linspace(0, 1, H).to(dev), torch.linspace(0, 1, W).to(dev))
                e.set_target(img)
                e.step([lr] * 5)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                e.step([lr] * 10)
                torch.cuda.synchronize(dev)
                dtf = (time.perf_counter() - t1) / 10
                out["by_scratch_format"][str(fmt)] = {"value": H * W / dtf / 1e6, "unit": "Mpixel-iters/s", "ms_per_step": dtf * 1e3, "steps": 10, "warmup": 5}
                e.close()
                del e
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.hidden, args.depth, args.cpu_size)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
