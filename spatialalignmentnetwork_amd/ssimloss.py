"""Host-side mirror of the reference's ssimloss.py on the fused HIP window kernel.
Reference: ssimloss.py:11-40."""
import torch

from . import ops


def ssimloss(X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    assert not torch.is_complex(X)
    assert not torch.is_complex(Y)
    return ops.ssim_loss(X.contiguous(), Y.contiguous())
