"""`make fit` entry: the reference's fit loop (implicit_image/compress.py:54-170) on the HIP engine.

    python -m implicit_image.fit [key=value ...]         # same override grammar as the Hydra entry

Reads conf/ (implicit_image/config.py), builds the registry model, fits with train_epoch /
update_connections / eval_epoch exactly in the reference's order, logs loss / PSNR / PSNR_8bit,
and saves `model.pth` (fp32 state_dict with the reference's parameter names) in the run directory.
Comma sweeps expand to a job list; under torch.distributed.run the jobs are sharded over ranks
(per-image sharding: one fit per GPU, no collectives).
"""
import copy
import json
import logging
import os
import sys
import time

import torch

from .config import Cfg, expand_sweeps, load_config
from .data import get_grid, load_img
from .models import registry as model_registry
from .parallel import shard_jobs
from .pipeline import entropy_coding
from .pipeline.quant import Quantize
from .utils.train_helper import (eval_epoch, get_device, get_optimizer_lr_scheduler, setup_mask, train_epoch,
                                 train_steps)

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def fit_one(cfg: Cfg, device: torch.device, out_dir: str = None):
    """One fit; returns dict(loss, PSNR, PSNR_8bit, steps, seconds)."""
    torch.manual_seed(cfg.seed)                                                   # compress.py:58
    img = load_img(**cfg.img)                                                      # compress.py:64
    grid = get_grid(cfg.img.height, cfg.img.width)                                 # compress.py:67
    small = cfg.masking.density if (cfg.get("masking") and cfg.masking.get("name") == "Small_Dense") else 1.0
    eng_kw = dict(cfg.get("engine") or {})
    model = model_registry[cfg.mlp.name](**cfg.mlp, small_dense_density=small, **eng_kw)   # compress.py:77
    model, grid, img = model.to(device), grid.to(device), img.to(device)          # compress.py:84-86
    model.train()
    optim, lr_scheduler = get_optimizer_lr_scheduler(model, cfg.optim)             # compress.py:105
    mult = cfg.train.multiplier
    num_steps = cfg.train.num_steps * mult                                         # compress.py:110-120
    mcfg = copy.deepcopy(cfg.get("masking")) if cfg.get("masking") else None
    if mcfg:
        if mcfg.get("end_when"):
            mcfg.end_when = int(mcfg.end_when * mult)
        if mcfg.get("interval"):
            mcfg.interval = int(mcfg.interval * mult)
    mask = setup_mask(model, optim, mcfg)                                          # compress.py:127
    t0, last = time.time(), {}
    i = -1
    while i + 1 < num_steps:                                                       # compress.py:137-170
        # the iterations up to the next one that needs the host (topology update / logging) go to the engine
        # in one call; train_steps() is bit-identical to calling train_epoch() once per iteration
        first = i + 1
        i = first
        while not ((mask and i <= mcfg.end_when and i % mcfg.interval == 0)
                   or (i + 1) % cfg.train.log_steps == 0 or i + 1 == num_steps):
            i += 1
        train_steps(model, optim, grid, img, i - first + 1, lr_scheduler=lr_scheduler, mask=mask)
        if mask and i <= mcfg.end_when and i % mcfg.interval == 0:
            mask.update_connections()
        if (i + 1) % cfg.train.log_steps == 0 or i + 1 == num_steps:
            _, loss, psnr, psnr8 = eval_epoch(model, grid, img)
            last = {"loss": loss, "PSNR": psnr, "PSNR_8bit": psnr8}
            msg = f"Train | Step: {i + 1} | " + " | ".join(f"{k}: {v:.4f}" for k, v in last.items())
            if mask:
                msg += f" | Prune Rate: {mask.prune_rate:.4f} | Density: {mask.stats.total_density:.4f}"
            logging.info(msg)
    last.update(steps=num_steps, seconds=time.time() - t0)
    quantized_model = None
    if cfg.get("quant"):                                                           # compress.py:172-240
        qcfg = copy.deepcopy(cfg.quant)
        depth = cfg.mlp.depth
        qcfg.skip_ll = [s.replace("layers.first.", "layers.0.").replace("layers.last.", f"layers.{depth - 1}.")
                        for s in (qcfg.get("skip_ll") or [])]
        quantized_model = copy.deepcopy(model)
        q_optim, q_sched = get_optimizer_lr_scheduler(quantized_model, cfg.optim, quantize_mode=True)
        # NOTE: the reference passes the ORIGINAL model's mask here, which makes the quant phase step the
        # wrong optimiser: its quantised copy is never updated and stays sparse.
        # This entry does fine-tune the copy, with the final topology held fixed: the copy's engine gets the
        # mask, so k_adam keeps every pruned weight at exactly 0 and find_centroids' label 0 stays the pruned set
        # (Quant PSNR, Compressed Bytes and Density then describe the artefact that is saved).
        if mask is not None:
            quantized_model.engine(grid, img)
            quantized_model.set_engine_masks(mask.flat_mask())
        with Quantize(quantized_model, q_optim, qcfg) as q:
            for i in range(qcfg.num_steps):
                train_epoch(quantized_model, q_optim, grid, img, lr_scheduler=q_sched)
                if (i + 1) % qcfg.log_steps == 0:
                    _, l, p, p8 = eval_epoch(quantized_model, grid, img)
                    logging.info(f"Quant | Step: {i + 1} | loss: {l:.4f} | PSNR: {p:.4f} | PSNR_8bit: {p8:.4f}")
        quantized_model = q.convert()
        _, l, p, p8 = eval_epoch(quantized_model, grid, img)
        last.update({"Quant loss": l, "Quant PSNR": p, "Quant PSNR 8bit": p8})
        if mask is not None:      # what fraction of the masked layers' weights the saved artefact really holds at zero
            zeros = sum(int((w == 0).sum().item()) for n, w in quantized_model.named_parameters() if n in mask.mask_dict)
            total = sum(w.numel() for n, w in quantized_model.named_parameters() if n in mask.mask_dict)
            last["Quant zero fraction"] = zeros / max(total, 1)
            last["Density"] = float(mask.stats.total_density)
        logging.info(f"Post Quant | Train step: {num_steps} | Quant step: {qcfg.num_steps} | Quant PSNR: {p:.4f}")
    if out_dir and cfg.train.save_weights:
        os.makedirs(out_dir, exist_ok=True)
        torch.save({"state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}},
                   os.path.join(out_dir, "model.pth"))                            # compress.py:243-244
        if quantized_model is not None and cfg.get("entropy_coding"):              # compress.py:249-263
            ec = dict(cfg.entropy_coding)
            nbytes = entropy_coding.compress_state_dict(quantized_model.half(), os.path.join(out_dir, "model_quantized"),
                                                        **ec)
            last["Compressed Bytes"] = nbytes
            logging.info(f"Compressed bytes {nbytes}")
        with open(os.path.join(out_dir, "result.json"), "w") as f:
            json.dump(last, f)
    return last


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(message)s")
    conf_dir = os.environ.get("IIC_CONF", os.path.join(REPO, "conf"))
    jobs = expand_sweeps(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    results = []
    for ov in shard_jobs(jobs, world, rank):
        cfg = load_config(conf_dir, ov)
        device = get_device(cfg.device)
        if device.type == "cuda":
            device = torch.device("cuda", local_rank)
            torch.cuda.set_device(device)
        tag = ",".join(o for o in ov if not o.startswith(("exp_name", "img.name"))) or "default"
        out_dir = os.path.join("outputs", str(cfg.img.name), str(cfg.exp_name), tag.replace("/", "_"))
        res = fit_one(cfg, device, out_dir)
        logging.info(f"[rank {rank}] {tag}: {res}")
        results.append((tag, res))
    return results


if __name__ == "__main__":
    main()
