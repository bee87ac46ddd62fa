"""Child of tests/test_gpu_host_state.py::test_pixel_split_two_real_engines: one rank of a two-rank pixel-split fit with a
REAL SirenEngine (row shard) on cuda:0, gloo collectives (RCCL refuses two ranks on one device).  RANK / WORLD_SIZE /
MASTER_* come from the environment and are set before anything touches the GPU.  Writes {losses, params sha256, grads}."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "implicit-image-compression_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from implicit_image._engine import SirenEngine
    from implicit_image.parallel import PixelSplitFit, shard_rows
    from oracle import siren_oracle as so           # (test infrastructure: the seed-0 init and the image formula)
    H, W, hidden, depth = 64, 48, 64, 4
    p = so.siren_init(hidden, depth, seed=0)
    img = so.synthetic_image(H, W, seed=5)
    r0, r1 = shard_rows(H, world, rank)
    eng = SirenEngine(H, W, hidden, depth, compute_dtype="f16", row_begin=r0, row_end=r1)
    gh, gw = so.grid_vectors(H, W)
    eng.set_coords(gh.cuda(), gw.cuda())
    eng.set_params(torch.tensor(so.flatten(p)).cuda())
    eng.set_target(img[r0:r1].contiguous().cuda())
    fit = PixelSplitFit(eng, 3 * H * W)
    losses = [fit.step(3e-4) for _ in range(5)]
    params = eng.get_params().cpu().numpy()
    json.dump({"rank": rank, "rows": [r0, r1], "losses": losses, "params_sha256": hashlib.sha256(params.tobytes()).hexdigest(),
               "params_head": params[:8].tolist()}, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
