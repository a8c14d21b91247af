"""Host-side mirror of the reference's lnccloss.py on the fused HIP window kernels.
Reference: lnccloss.py:7-65.  Both losses are differentiable wrt both images: lncc_loss through
san_lncc_loss_bwd, ms_lncc_loss additionally through the adjoint of the Gaussian + average-pool
down-sampler (san_smooth_pool_bwd); see autograd.py."""
import torch

from . import autograd


def lncc_loss(I: torch.Tensor, J: torch.Tensor, win=None) -> torch.Tensor:
    ndims = len(I.shape) - 2
    assert ndims == 2, "volumes should be 2 dimensions. found: %d" % ndims
    if win is None:
        win = [9] * ndims
    assert win[0] == win[1]
    return autograd.lncc_loss(I, J, int(win[0]))


_GAUSS = {}


def _gaussian_kernel_2d(sigma: float, device) -> torch.Tensor:
    """Outer product of two normalised 1-D Gaussians with 2*ceil(2*sigma)+1 taps, renormalised
    (miloss.py:6-18); built once on the host and cached on the device."""
    key = (float(sigma), str(device))
    if key not in _GAUSS:
        import math
        size = int(2 * math.ceil(sigma * 2) + 1)
        x = torch.linspace(-(size - 1) // 2, (size - 1) // 2, size)
        g = 1.0 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-(x ** 2) / (2 * sigma ** 2))
        g = g / torch.sum(g)
        k = torch.tensordot(g, g, 0)
        _GAUSS[key] = (k / torch.sum(k)).contiguous().to(device)
    return _GAUSS[key]


def ms_lncc_loss(I: torch.Tensor, J: torch.Tensor, win=None, ms=3, sigma=3) -> torch.Tensor:
    """Multi-scale LNCC: LNCC at `ms` scales, Gaussian smoothing + 2x average pooling in
    between, averaged.  lnccloss.py:58-65."""
    k = _gaussian_kernel_2d(sigma, I.device)
    loss = lncc_loss(I, J, win)
    for _ in range(ms - 1):
        I, J = autograd.smooth_pool(I, k), autograd.smooth_pool(J, k)
        loss = loss + lncc_loss(I, J, win)
    return loss / ms
