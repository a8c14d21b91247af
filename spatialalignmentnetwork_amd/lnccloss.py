"""Host-side mirror of the reference's lnccloss.py on the fused HIP window kernel.
Reference: lnccloss.py:7-65."""
import torch

from . import ops


def lncc_loss(I: torch.Tensor, J: torch.Tensor, win=None) -> torch.Tensor:
    ndims = len(I.shape) - 2
    assert ndims == 2, "volumes should be 2 dimensions. found: %d" % ndims
    if win is None:
        win = [9] * ndims
    assert win[0] == win[1]
    return ops.lncc_loss(I.contiguous(), J.contiguous(), int(win[0]))
