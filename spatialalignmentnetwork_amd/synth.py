"""Deterministic synthetic inputs and weights (host side, numpy/torch CPU).

There is no network for datasets or checkpoints, so benchmarks, parity tests and
golden fixtures all draw from the generators below.  Everything is keyed by
name + seed through numpy's Philox counter RNG, so the same call gives the same
bits in the build container, on the GPU box and inside
``tests/golden/make_golden.py``.

Input statistics follow SURVEY.md section 8(d): DICOM-derived slices are real
magnitudes normalised to [0, 1] (reference: convert_fastMRIDICOM.py:13-16,
paired_dataset.py:69-73); the auxiliary contrast is a remapped copy displaced by
a rigid + B-spline field with the reference's own augmentation ranges
(augment.py:10-11,40-48).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _rng(name: str, seed: int) -> np.random.Generator:
    key = (zlib.crc32(name.encode()) << 32) | (seed & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def fill_params(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0,
                dtype=torch.float32, damp: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic non-trivial values for every state_dict entry.

    ``damp`` < 1 scales the output convolution (``up_conv.<last>.1``) of every CASCADE regulariser: each cascade then
    applies a small correction, as in a trained network, instead of the O(1) random map of the default weights (whose
    12-fold composition amplifies fp32 rounding noise by 3-4 orders of magnitude: the reference's own fp32 and fp64
    gradients differ by 16 % there).  Used by the tight full-size gradient fixtures.

    Conv weights ~ U(-b, b), b = 1/sqrt(fan_in) (PyTorch's default bound, so
    activations stay O(1) through 12 cascades); BatchNorm affine/running stats,
    ``dc_weight`` and the alignment head are made non-trivial so parity tests
    exercise them (the reference zero-initialises the head: cross.py:20-21).
    """
    out: Dict[str, torch.Tensor] = {}
    for name, shape in named_shapes:
        shape = tuple(int(s) for s in shape)
        g = _rng(name, seed)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros((), dtype=torch.int64)
            continue
        if leaf == "dc_weight":
            v = g.uniform(0.5, 1.5, shape)
        elif leaf == "running_var":
            v = g.uniform(0.5, 1.5, shape)
        elif leaf == "running_mean":
            v = g.uniform(-0.1, 0.1, shape)
        elif len(shape) == 4:
            # Conv2d [Cout,Cin,kh,kw] / ConvTranspose2d [Cin,Cout,kh,kw]
            if "up_transpose_conv" in name:
                fan_in = shape[0]          # each output pixel sees Cin inputs (2x2 s2)
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            b = 1.0 / math.sqrt(fan_in)
            if name.endswith("net.2.weight"):
                b *= 0.02                  # alignment head: offsets of a few pixels
            v = g.uniform(-b, b, shape)
        elif len(shape) == 1 and leaf == "weight":
            v = g.uniform(0.5, 1.5, shape)  # BatchNorm gamma (mask weights too)
        elif len(shape) == 1 and leaf == "bias":
            b = 0.005 if name.endswith("net.2.bias") else 0.1
            v = g.uniform(-b, b, shape)
        else:
            v = g.uniform(-0.1, 0.1, shape)
        if damp != 1.0 and name.startswith("cascades.") and ".up_conv." in name and name.rsplit(".", 2)[-2] == "1":
            v = v * damp
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float64)).to(dtype)
    return out


def equispaced_pruned(shape: int, sparsity: float, start: int = 0) -> torch.Tensor:
    """Bool [shape], True = column NOT sampled.  Same construction as the
    reference's EquispacedMask (masks.py:86-110) with the random start fixed:
    a fully sampled centre of round(shape*sparsity*0.32) lines that lives at
    the two borders of the un-shifted axis, plus equispaced lines elsewhere."""
    center = round(shape * sparsity * 0.32)
    pruned = torch.zeros(shape, dtype=torch.bool)
    lo, hi = center // 2, center // 2 - center        # hi is negative
    pruned[lo:hi] = True
    remaining = math.floor(sparsity * shape - center)
    interval = int((shape - center - 1) // (remaining - 1))
    start_max = (shape - center) - ((remaining - 1) * interval + 1)
    assert 0 <= start <= start_max
    part = pruned[lo:hi].clone()
    part = torch.roll(part, part.shape[0] // 2)
    part[start:start + interval * remaining:interval] = False
    part = torch.roll(part, (part.shape[0] + 1) // 2)
    pruned[lo:hi] = part
    return pruned


def _blobs(h: int, w: int, g: np.random.Generator, nblobs: int = 12) -> np.ndarray:
    yy, xx = np.meshgrid(np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    a, b = g.uniform(0.6, 0.85), g.uniform(0.7, 0.9)
    img = 0.35 * (((xx / a) ** 2 + (yy / b) ** 2) < 1.0).astype(np.float64)
    # soft edge so the phantom is band-limited-ish
    img = img * np.clip((1.0 - ((xx / a) ** 2 + (yy / b) ** 2)) * 8.0, 0.0, 1.0)
    for _ in range(nblobs):
        cx, cy = g.uniform(-0.55, 0.55, 2)
        s = g.uniform(0.05, 0.22)
        amp = g.uniform(0.1, 0.5)
        img += amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    img = np.clip(img, 0.0, None)
    return img / max(img.max(), 1e-12)


def phantom_pair(n: int, c: int, h: int, w: int, seed: int = 1234) -> Tuple[torch.Tensor, torch.Tensor]:
    """(img_full, img_aux) complex64 [n, c, h, w].

    img_full: smooth phantom in [0, 1], imag = 0 for c == 1; for c > 1 it is
    multiplied by c smooth complex coil maps.  img_aux: contrast remapped
    (1 - x^0.7 inside the head) and displaced by rigid (rot <= 2*pi*0.005 rad,
    translation <= 0.05) + B-spline (9x9 control grid, +-1/50, bicubic) fields.
    """
    full = np.zeros((n, 1, h, w), dtype=np.float64)
    for i in range(n):
        full[i, 0] = _blobs(h, w, _rng(f"phantom{i}", seed))
    full_t = torch.from_numpy(full).float()
    aux_t = torch.where(full_t > 0.02, 1.0 - full_t.clamp(0, 1) ** 0.7, torch.zeros_like(full_t))

    g = _rng("deform", seed)
    rot = g.uniform(-2 * math.pi * 0.005, 2 * math.pi * 0.005, n)
    tr = g.uniform(-0.05, 0.05, n)
    theta = torch.zeros(n, 2, 3)
    for i in range(n):
        cr, sr = math.cos(rot[i]), math.sin(rot[i])
        theta[i] = torch.tensor([[cr, -sr, tr[i]], [sr, cr, tr[i]]])
    grid = F.affine_grid(theta, (n, 1, h, w), align_corners=False)
    ctl = torch.from_numpy((g.uniform(0, 1, (n, 2, 9, 9)) - 0.5) * 2 / 50).float()
    ctl = F.interpolate(ctl, size=(h, w), mode="bicubic", align_corners=False)
    grid = grid + ctl.permute(0, 2, 3, 1)
    aux_t = F.grid_sample(aux_t, grid, mode="bilinear", padding_mode="reflection", align_corners=False)

    if c == 1:
        img_full = torch.complex(full_t, torch.zeros_like(full_t))
        img_aux = torch.complex(aux_t, torch.zeros_like(aux_t))
    else:
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
        maps = []
        for k in range(c):
            ang = 2 * math.pi * k / c
            cx, cy = 1.2 * math.cos(ang), 1.2 * math.sin(ang)
            mag = torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 2.0)
            ph = 0.8 * (xx * math.cos(ang) + yy * math.sin(ang))
            maps.append(torch.polar(mag, ph))
        m = torch.stack(maps)[None]                       # [1,c,h,w] complex
        m = m / (m.abs() ** 2).sum(1, keepdim=True).sqrt()
        img_full = full_t.to(torch.complex64) * m
        img_aux = aux_t.to(torch.complex64) * m
    return img_full.contiguous(), img_aux.contiguous()


def random_kspace(n: int, c: int, h: int, w: int, seed: int = 7) -> torch.Tensor:
    """The reference micro-benchmark's own input distribution (model.py:372-375):
    uniform [0,1) real and imaginary parts."""
    g = _rng("rand_kspace", seed)
    re = torch.from_numpy(g.uniform(0, 1, (n, c, h, w))).float()
    im = torch.from_numpy(g.uniform(0, 1, (n, c, h, w))).float()
    return torch.complex(re, im)
