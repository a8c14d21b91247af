"""Host-side mirror of the reference's signal_utils.py (fft2/ifft2/rss and the
shift helpers), executing on libsan_hip.so.  Reference: signal_utils.py:4-26.
fft2 / ifft2 / rss are differentiable (autograd.py: hand-written adjoints on the same kernels)."""
import torch

from . import autograd, ops  # noqa: F401


def fft2(x: torch.Tensor) -> torch.Tensor:
    """Orthonormal 2-D FFT over the last two axes, DC at index 0.  signal_utils.py:4-7."""
    assert len(x.shape) == 4
    return autograd.fft2c(x, inverse=False)


def ifft2(x: torch.Tensor) -> torch.Tensor:
    """Orthonormal inverse 2-D FFT.  signal_utils.py:9-12."""
    assert len(x.shape) == 4
    return autograd.fft2c(x, inverse=True)


def fftshift2(x: torch.Tensor) -> torch.Tensor:
    """signal_utils.py:14-17 (visualisation only; a memory roll, no arithmetic)."""
    assert len(x.shape) == 4
    return torch.roll(x, (x.shape[-2] // 2, x.shape[-1] // 2), dims=(-2, -1))


def ifftshift2(x: torch.Tensor) -> torch.Tensor:
    """signal_utils.py:19-22."""
    assert len(x.shape) == 4
    return torch.roll(x, ((x.shape[-2] + 1) // 2, (x.shape[-1] + 1) // 2), dims=(-2, -1))


def rss(x: torch.Tensor) -> torch.Tensor:
    """Root-sum-of-squares over dim 1 (complex aware), keepdim.  signal_utils.py:24-26."""
    assert len(x.shape) == 4
    return autograd.rss(x)
