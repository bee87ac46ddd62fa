import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-image-compression_amd"))
from implicit_image.data import get_grid
from implicit_image.models import registry
from implicit_image.utils.train_helper import eval_epoch, get_optimizer_lr_scheduler, setup_mask, train_epoch
d = np.load(os.path.join(ROOT, "tests/golden/rigl_256x8_48.npz"))
class Cfg(dict):
    __getattr__ = dict.get
for fmt in (16, 12):
    torch.manual_seed(0)
    model = registry["siren"](depth=8, hidden_size=256, first_omega_0=50, hidden_omega_0=30, scratch_format=fmt).to("cuda")
    optim, sched = get_optimizer_lr_scheduler(model, Cfg(name="adam", lr=3e-4))
    mcfg = Cfg(name="RigL", density=0.1, sparse_init="erdos-renyi-kernel", dense_gradients=True, growth_mode="absolute-gradient",
               prune_mode="magnitude", redistribution_mode="none", dense=False, prune_rate=0.1, decay_schedule="cosine", end_when=90, interval=20)
    mask = setup_mask(model, optim, mcfg)
    img, grid = torch.tensor(d["img"]).cuda(), get_grid(48, 48).cuda()
    ls = []
    for i in range(100):
        ls.append(train_epoch(model, optim, grid, img, lr_scheduler=sched, mask=mask))
        if i <= 90 and i % 20 == 0:
            mask.update_connections()
    _, _, psnr, _ = eval_epoch(model, grid, img)
    r = np.array(ls) / d["losses"]
    print(f"fmt {fmt}: psnr {psnr:.4f} ref {float(d['psnr']):.4f}; loss ratio at 0,1,2,5,10,19,20,21,25,40,60,80,99:", np.round(r[[0,1,2,5,10,19,20,21,25,40,60,80,99]], 4))
    print("   ref losses", np.round(d["losses"][[0,1,20,21,40,60,80,99]], 6))
