"""Validation metrics with the reference's names and conventions (metrics.py:23-69), computed on the GPU:
mse, mae, nmse, psnr (data_range 1, over the whole batch like skimage's peak_signal_noise_ratio), ssim
(7x7 uniform window, data_range 1, sample covariance = 1 - ssimloss) and mi (64 x 64 joint histogram)."""
from __future__ import annotations

import math

import torch

from . import ops


def _sums(gt: torch.Tensor, pred: torch.Tensor, bins: int = 64) -> torch.Tensor:
    assert gt.shape == pred.shape and gt.dim() == 4, "wrong shape [batch, channel=1, rows, cols]"
    return ops.image_metrics(gt.contiguous(), pred.contiguous(), bins).cpu()     # [n*c, 4] float64


def mse(gt, pred) -> float:
    s = _sums(gt, pred)
    return (s[:, 0].sum() / gt.numel()).item()


def mae(gt, pred) -> float:
    s = _sums(gt, pred)
    return (s[:, 1].sum() / gt.numel()).item()


def nmse(gt, pred) -> float:
    s = _sums(gt, pred)
    return (s[:, 0].sum() / s[:, 2].sum()).item()


def psnr(gt, pred) -> float:
    m = mse(gt, pred)
    return 10.0 * math.log10(1.0 / m) if m > 0 else float("inf")


def ssim(gt, pred) -> float:
    """Mean structural similarity of the batch = 1 - ssimloss (same window, constants and covariance norm)."""
    return 1.0 - ops.ssim_loss(gt.contiguous(), pred.contiguous()).item()


def mi(gt, pred, bins: int = 64) -> float:
    assert gt.shape == pred.shape
    s = _sums(gt, pred, bins)
    return s[:, 3].mean().item()


def all_metrics(gt, pred) -> dict:
    """One pass for everything CSModel.test() reports."""
    s = _sums(gt, pred)
    m = (s[:, 0].sum() / gt.numel()).item()
    return {"MSE": m, "MAE": (s[:, 1].sum() / gt.numel()).item(),
            "PSNR": 10.0 * math.log10(1.0 / m) if m > 0 else float("inf"), "MI": s[:, 3].mean().item()}


def test_metrics(gt, pred, warped) -> dict:
    """Every scalar CSModel.test() reports (model.py:275-279) from ONE host synchronisation: the sums of (gt, pred), the
    mutual information of (gt, warped) and the SSIM loss of (gt, pred) are computed on the device and cross the bus together."""
    assert gt.shape == pred.shape == warped.shape and gt.dim() == 4, "wrong shape [batch, channel=1, rows, cols]"
    a = ops.image_metrics(gt.contiguous(), pred.contiguous(), 64)          # [n*c, 4] float64
    b = ops.image_metrics(gt.contiguous(), warped.contiguous(), 64)
    ssl = ops.ssim_loss(gt.contiguous(), pred.contiguous())
    host = torch.cat([a.reshape(-1), b.reshape(-1), ssl.double().reshape(1)]).cpu()      # the one synchronisation
    k = a.numel()
    a, b, ssl = host[:k].reshape(-1, 4), host[k:2 * k].reshape(-1, 4), host[2 * k].item()
    m = (a[:, 0].sum() / gt.numel()).item()
    return {"MSE": m, "MAE": (a[:, 1].sum() / gt.numel()).item(), "PSNR": 10.0 * math.log10(1.0 / m) if m > 0 else float("inf"),
            "SSIM": 1.0 - ssl, "MI": b[:, 3].mean().item()}
