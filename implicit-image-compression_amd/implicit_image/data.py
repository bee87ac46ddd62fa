"""Coordinate grid (reference: implicit_image/data.py:78-88).

The engine never materialises the [H, W, 2] grid: it indexes the two 1-D `torch.linspace` vectors
per pixel.  `get_grid` still returns the reference's tensor so callers written
against the reference keep working; `grid_vectors` recovers the two vectors from such a grid.
"""
from typing import Tuple

import torch


def get_grid(height: int, width: int, device: torch.device = torch.device("cpu")) -> torch.Tensor:
    """[H, W, 2] pixel coordinates in the unit square, row coordinate first ('ij' order)."""
    rows = torch.linspace(0, 1, height, device=device)
    cols = torch.linspace(0, 1, width, device=device)
    return torch.stack((rows[:, None].expand(height, width), cols[None, :].expand(height, width)), dim=-1)


def grid_vectors(grid: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Row / column coordinate vectors of a separable [H, W, 2] grid; raises if the grid is not the
    outer product layout `get_grid` produces (the engine cannot represent an arbitrary grid)."""
    if grid.dim() != 3 or grid.shape[-1] != 2:
        raise ValueError(f"expected a [H, W, 2] grid, got {tuple(grid.shape)}")
    rows, cols = grid[:, 0, 0].contiguous(), grid[0, :, 1].contiguous()
    if not (torch.equal(grid[..., 0], rows[:, None].expand(grid.shape[:2]))
            and torch.equal(grid[..., 1], cols[None, :].expand(grid.shape[:2]))):
        raise ValueError("grid is not separable (rows x cols); only get_grid()-style grids are supported")
    return rows, cols


# ---------------------------------------------------------------------------------------------
# Image loading ("next" row §8f-3; reference data.py:44-75).  cv2 / kornia are not available
# offline, so this is an own P6 reader and torch resampling: PARITY UNPINNED for `resize-crop`
# (kornia's version-dependent bilinear resize); centre-crop of an integer window is exact.
# ---------------------------------------------------------------------------------------------
def synthetic_image(height: int, width: int, seed: int = 1234, device="cpu") -> torch.Tensor:
    """SURVEY.md §8(d) formula image: sinusoids + seeded uniform noise (amplitude 0.05), in [0,1]."""
    ys = torch.linspace(0, 1, height)[:, None].expand(height, width)
    xs = torch.linspace(0, 1, width)[None, :].expand(height, width)
    kx = torch.tensor([1.0, 2.0, 3.0])
    ky = torch.tensor([3.0, 1.0, 2.0])
    img = 0.5 + 0.25 * torch.sin(12 * xs[..., None] * kx) + 0.25 * torch.cos(9 * ys[..., None] * ky)
    g = torch.Generator().manual_seed(seed)
    img = img + 0.05 * (torch.rand(height, width, 3, generator=g) * 2 - 1)
    return img.clamp(0, 1).float().contiguous().to(device)


def read_ppm(path: str) -> torch.Tensor:
    """Binary P6 PPM (8 or 16 bit, big-endian samples) -> [H, W, 3] integer tensor (int32)."""
    import numpy as np
    with open(path, "rb") as f:
        data = f.read()
    tokens, pos = [], 0
    while len(tokens) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tokens.append(data[pos:end])
        pos = end
    if tokens[0] != b"P6":
        raise ValueError(f"{path}: not a binary PPM (P6)")
    w, h, maxval = int(tokens[1]), int(tokens[2]), int(tokens[3])
    pos += 1                                               # single whitespace after maxval
    dt = np.dtype(">u2") if maxval > 255 else np.dtype("u1")
    arr = np.frombuffer(data, dtype=dt, count=w * h * 3, offset=pos).reshape(h, w, 3)
    return torch.from_numpy(arr.astype(np.int32))


def load_img(path: str, height: int = 256, width: int = 256, bits: int = 8, plot: bool = False,
             crop_mode: str = "centre-crop", save_gt: bool = False, seed: int = 1234, **kwargs) -> torch.Tensor:
    """[H, W, 3] float32 image in [0,1] (reference signature).  `path` = "synthetic" (or
    "synthetic:<seed>") generates the formula image at (height, width)."""
    if path.startswith("synthetic"):
        if ":" in path:
            seed = int(path.split(":", 1)[1])
        return synthetic_image(height, width, seed)
    raw = read_ppm(path)
    img = (raw.double() / (2 ** bits - 1)).float().permute(2, 0, 1)          # data.py:54-55
    if crop_mode == "resize-crop":
        smaller = min(height, width)
        _, h, w = img.shape
        scale = smaller / min(h, w)
        nh, nw = max(height, round(h * scale)), max(width, round(w * scale))
        img = torch.nn.functional.interpolate(img[None], size=(nh, nw), mode="bilinear", align_corners=False)[0]
    _, h, w = img.shape
    if h < height or w < width:
        raise ValueError(f"image {h}x{w} is smaller than the requested {height}x{width} crop")
    top, left = (h - height) // 2, (w - width) // 2
    return img[:, top:top + height, left:left + width].permute(1, 2, 0).contiguous()
