"""On-device counterpart of the reference's training augmentation (augment.py:7-66): a random rigid
motion (rotation <= 2*pi*0.005 rad about the image centre, translation <= 0.05 on both axes) plus an
optional random B-spline deformation (9 x 9 control offsets in +-1/50, bicubic up-sampled), applied by
bilinear resampling with reflection padding.  Same function names, arguments and return values as the
reference; the grid and the resampling run in HIP kernels (san_augment_grid, san_grid_sample_*_fwd)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops

ROTATION = 2 * np.pi * 0.005        # augment.py:10
TRANSLATION = 0.05                  # augment.py:11
BSPLINE_SCALE = 50                  # augment.py:42
BSPLINE_POINTS = 9                  # augment.py:43


def rigid_affine(r_s, t_s, device) -> torch.Tensor:
    """[N,2,3] matrices T @ R built in float64 on the host (augment.py:15-33), float32 on the device."""
    mats = []
    for r, t in zip(r_s, t_s):
        rot = np.array([[np.cos(r), -np.sin(r), 0.0], [np.sin(r), np.cos(r), 0.0], [0.0, 0.0, 1.0]])
        tr = np.array([[1.0, 0.0, t], [0.0, 1.0, t], [0.0, 0.0, 1.0]])
        mats.append((tr @ rot)[:-1])
    return torch.as_tensor(np.stack(mats, 0), dtype=torch.float32).to(device).contiguous()


def rigid_grid(img: torch.Tensor) -> torch.Tensor:
    """Random rigid sampling grid [N,H,W,2] (same draws as the reference: two np.random.uniform calls)."""
    n = img.shape[0]
    r_s = np.random.uniform(-ROTATION, ROTATION, n)
    t_s = np.random.uniform(-TRANSLATION, TRANSLATION, n)
    return ops.augment_grid(rigid_affine(r_s, t_s, img.device), None, img.shape[2], img.shape[3])


def bspline_ctrl(img: torch.Tensor) -> torch.Tensor:
    """(rand - 0.5) * 2 / scale control offsets [N,2,9,9] (augment.py:42-44)."""
    return ((torch.rand(img.shape[0], 2, BSPLINE_POINTS, BSPLINE_POINTS, device=img.device, dtype=torch.float32) - 0.5)
            * 2 / BSPLINE_SCALE).contiguous()


def bspline_grid(img: torch.Tensor) -> torch.Tensor:
    """Random B-spline offset field [N,H,W,2] (identity affine excluded, like the reference's bspline_grid)."""
    zero = torch.zeros((img.shape[0], 2, 3), device=img.device, dtype=torch.float32)
    return ops.augment_grid(zero, bspline_ctrl(img), img.shape[2], img.shape[3])


def sample(img: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """Bilinear, reflection padding, align_corners=False; complex images in one launch (augment.py:60-65)."""
    if torch.is_complex(img):
        return ops.grid_sample_complex(img.contiguous(), grid, padding="reflection")
    return ops.grid_sample(img.contiguous(), grid, padding="reflection")


def augment(img: torch.Tensor, rigid: bool = True, bspline: bool = True, grid: torch.Tensor = None):
    """Returns (augmented image, grid).  Either draws a new rigid (+ B-spline) grid or re-applies ``grid``
    (rigid=False, bspline=False), exactly like the reference (augment.py:50-66)."""
    if grid is None:
        assert rigid is True
        n = img.shape[0]
        r_s = np.random.uniform(-ROTATION, ROTATION, n)
        t_s = np.random.uniform(-TRANSLATION, TRANSLATION, n)
        ctrl = bspline_ctrl(img) if bspline else None
        grid = ops.augment_grid(rigid_affine(r_s, t_s, img.device), ctrl, img.shape[2], img.shape[3])
    else:
        assert rigid is False
        assert bspline is False
    return sample(img, grid), grid
