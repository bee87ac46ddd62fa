import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image.data import get_grid
from implicit_image.models import registry
from implicit_image.utils.train_helper import eval_epoch, get_optimizer_lr_scheduler, setup_mask, train_epoch
from implicit_image.pipeline.masking import funcs
d = np.load(os.path.join(ROOT, "tests/golden/rigl_256x8_48.npz"))
class Cfg(dict):
    __getattr__ = dict.get
torch.manual_seed(0)
model = registry["siren"](depth=8, hidden_size=256, first_omega_0=50, hidden_omega_0=30).to("cuda")
optim, sched = get_optimizer_lr_scheduler(model, Cfg(name="adam", lr=3e-4))
mcfg = Cfg(name="RigL", density=0.1, sparse_init="erdos-renyi-kernel", dense_gradients=True, growth_mode="absolute-gradient",
           prune_mode="magnitude", redistribution_mode="none", dense=False, prune_rate=0.1, decay_schedule="cosine", end_when=90, interval=20)
mask = setup_mask(model, optim, mcfg)
img, grid = torch.tensor(d["img"]).cuda(), get_grid(48, 48).cuda()
orig = funcs.abs_grad_growth
def dbg(masking, name, total_regrowth, weight):
    nm = masking.mask_dict[name].data.bool()
    if "layers.7" in name or "layers.0" in name:
        cand = (nm == 0)
        g = weight.grad[cand].abs()
        print(f"  grow {name}: zeros {int(cand.sum())} regrow {int(total_regrowth)} cand |g|>0: {int((g > 0).sum())} min {g.min().item():.3e} nan {int(torch.isnan(weight.grad).sum())}")
    if "layers.7" in name:
        cand = (nm == 0)
        z = ((weight.grad == 0) & cand).nonzero()
        for c, j in z.tolist():
            m6 = masking.mask_dict["layers.6.linear.weight"]
            w6 = dict(masking.module.named_parameters())["layers.6.linear.weight"]
            b6 = dict(masking.module.named_parameters())["layers.6.linear.bias"]
            print(f"   zero grad at c={c} j={j}: grad column j = {weight.grad[:, j].tolist()}  layer-6 row {j}: mask nnz {int(m6[j].sum())} |w| sum {w6.data[j].abs().sum().item():.3e} bias {b6.data[j].item():.4e} phase rev {b6.data[j].item()*30/6.283185307:.5f}")
    return orig(masking, name, total_regrowth, weight)
import implicit_image.pipeline.masking.core as core
for reg in (getattr(core, "grow_registry", None), getattr(funcs, "grow_registry", None)):
    if reg:
        for k, v in list(reg.items()):
            if v is orig: reg[k] = dbg
oprune = funcs.magnitude_prune
def dbgp(masking, mask, weight, name):
    out = oprune(masking, mask, weight, name)
    if "layers.7" in name:
        print(f"  prune {name}: rate {masking.name2prune_rate[name]:.4f} nnz_before {masking.stats.nonzeros_dict[name]} zeros {masking.stats.zeros_dict[name]} mask zeros after {int((out == 0).sum())} w==0: {int((weight.data == 0).sum())} |w| min {weight.data.abs().min().item():.3e}")
    return out
for reg in (getattr(core, "prune_registry", None), getattr(funcs, "prune_registry", None)):
    if reg:
        for k, v in list(reg.items()):
            if v is oprune: reg[k] = dbgp
for i in range(1):
    train_epoch(model, optim, grid, img, lr_scheduler=sched, mask=mask)
    if i <= 90 and i % 20 == 0:
        print("update at", i)
        mask.update_connections()
        print("  nnz", [int(mask.mask_dict[n].sum().item()) for n in mask.mask_dict])
