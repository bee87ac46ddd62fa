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
