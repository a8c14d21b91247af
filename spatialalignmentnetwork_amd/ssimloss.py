"""Host-side mirror of the reference's ssimloss.py on the fused HIP window kernel.
Reference: ssimloss.py:11-40."""
import torch

from . import autograd


def ssimloss(X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    assert not torch.is_complex(X)
    assert not torch.is_complex(Y)
    """1 - mean SSIM; differentiable wrt both images (autograd._SsimLossFn -> san_ssim_loss_bwd_dev)."""
    return autograd.ssimloss(X, Y)
