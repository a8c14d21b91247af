"""Host-side mirror of the reference's 1-D column masks (masks.py:7-125): an
nn.Module with a ``weight`` parameter and a bool ``pruned`` buffer (True = column
not sampled), sampled once per model on the host and stored in the checkpoint.
Low frequencies live at the two borders of the un-shifted axis."""
from __future__ import annotations

import math
import random

import torch

from .synth import equispaced_pruned


class Mask(torch.nn.Module):
    """masks.py:7-46 (the learned-pruning helpers are out of scope)."""

    def __init__(self, shape):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(shape))
        self.register_buffer("pruned", torch.zeros(shape, dtype=torch.bool))


class StandardMask(Mask):
    """Random lines outside a fully sampled centre of 0.32*sparsity.  masks.py:48-69."""

    def __init__(self, sparsity, shape):
        super().__init__(shape)
        center = round(shape * sparsity * 0.32)
        other = (sparsity * shape - center) / (shape - center)
        prob = torch.ones(shape) * 1.1
        prob[center // 2:center // 2 - center] = other
        _, ind = torch.topk(prob - torch.rand(shape), math.floor(sparsity * shape))
        pruned = torch.ones(shape, dtype=torch.bool)
        pruned[ind] = False
        self.pruned = pruned


class EquispacedMask(Mask):
    """Equispaced lines outside the centre, random start.  masks.py:86-110."""

    def __init__(self, sparsity, shape, start=None):
        super().__init__(shape)
        center = round(shape * sparsity * 0.32)
        remaining = math.floor(sparsity * shape - center)
        interval = int((shape - center - 1) // (remaining - 1))
        start_max = (shape - center) - ((remaining - 1) * interval + 1)
        if start is None:
            start = random.randint(0, start_max)
        self.pruned = equispaced_pruned(shape, sparsity, start)


class LowpassMask(Mask):
    """Centre only.  masks.py:112-125."""

    def __init__(self, sparsity, shape):
        super().__init__(shape)
        center = math.floor(shape * sparsity)
        pruned = torch.zeros(shape, dtype=torch.bool)
        pruned[center // 2:center // 2 - center] = True
        self.pruned = pruned


masks = {"mask": Mask, "standard": StandardMask, "lowpass": LowpassMask, "equispaced": EquispacedMask}
