"""CPU oracle for the reconstruction + alignment hot path.

TEST INFRASTRUCTURE ONLY.  This module is the checker, never the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``spatialalignmentnetwork_amd/`` imports it and the
product path fails loudly when the HIP library is missing.

What it is: a functional (no ``nn.Module``) restatement, in plain PyTorch CPU
ops, of the arithmetic the reference performs on the hot path.  The reference's
own arithmetic lives in PyTorch ATen (third-party, un-vendored, un-pinned: the
reference has no requirements file); this oracle therefore calls the same ATen
CPU kernels (``torch.fft``, ``F.conv2d`` ...) but composes them itself, driven
by a flat ``{state_dict key: tensor}`` parameter dictionary.

Pinning: the reference holds no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned instead against outputs of the
reference itself, produced in the build container by importing
``/root/reference`` (``tests/golden/make_golden.py``) and committed as ``.npz``
fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every
function below against them.

Every function cites the reference file:line whose behaviour it restates.
All functions are dtype-generic (float32 for parity, float64 as arbiter).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------
# signal primitives
# --------------------------------------------------------------------------


def fft2(x: torch.Tensor) -> torch.Tensor:
    """Orthonormal 2-D DFT over the last two axes, DC at index 0 (no shift).
    Reference: signal_utils.py:4-7."""
    assert x.dim() == 4
    return torch.fft.fftn(x, dim=(-2, -1), norm="ortho")


def ifft2(x: torch.Tensor) -> torch.Tensor:
    """Orthonormal inverse 2-D DFT.  Reference: signal_utils.py:9-12."""
    assert x.dim() == 4
    return torch.fft.ifftn(x, dim=(-2, -1), norm="ortho")


def rss(x: torch.Tensor) -> torch.Tensor:
    """Root-sum-of-squares over the coil axis (complex aware), keepdim.
    Reference: signal_utils.py:24-26."""
    assert x.dim() == 4
    if torch.is_complex(x):
        return (x.real * x.real + x.imag * x.imag).sum(dim=1, keepdim=True).sqrt()
    return (x * x).sum(dim=1, keepdim=True).sqrt()


def fftshift2(x: torch.Tensor) -> torch.Tensor:
    """Reference: signal_utils.py:14-17."""
    return torch.roll(x, (x.shape[-2] // 2, x.shape[-1] // 2), dims=(-2, -1))


# --------------------------------------------------------------------------
# cascade U-Net (fastMRI style: conv3x3 no-bias -> InstanceNorm -> LeakyReLU 0.2)
# --------------------------------------------------------------------------


def _inorm(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """InstanceNorm2d(affine=False): biased variance, eps inside the sqrt.
    Reference: varnet.py:141,144,180,235."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def conv_block(x: torch.Tensor, w_a: torch.Tensor, w_b: torch.Tensor) -> torch.Tensor:
    """Two (conv3x3 pad1 no-bias, InstanceNorm, LeakyReLU 0.2) stages.
    Reference: varnet.py:139-146."""
    x = F.leaky_relu(_inorm(F.conv2d(x, w_a, padding=1)), 0.2)
    x = F.leaky_relu(_inorm(F.conv2d(x, w_b, padding=1)), 0.2)
    return x


def transpose_conv_block(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d 2x2 stride 2 no-bias, InstanceNorm, LeakyReLU 0.2.
    Reference: varnet.py:176-182.  ``w`` is [Cin, Cout, 2, 2]."""
    return F.leaky_relu(_inorm(F.conv_transpose2d(x, w, stride=2)), 0.2)


def unet_forward(p: Params, pre: str, x: torch.Tensor, num_pools: int) -> torch.Tensor:
    """fastMRI U-Net.  Reference: varnet.py:82-119 (forward), :60-80 (layout).
    ``pre`` is the state_dict prefix of the Unet (ending in '.')."""
    stack = []
    out = x
    for i in range(num_pools):
        out = conv_block(out, p[f"{pre}down_sample_layers.{i}.layers.0.weight"],
                         p[f"{pre}down_sample_layers.{i}.layers.3.weight"])
        stack.append(out)
        out = F.avg_pool2d(out, kernel_size=2, stride=2)
    out = conv_block(out, p[f"{pre}conv.layers.0.weight"], p[f"{pre}conv.layers.3.weight"])
    for i in range(num_pools):
        skip = stack.pop()
        out = transpose_conv_block(out, p[f"{pre}up_transpose_conv.{i}.layers.0.weight"])
        pad_r = 1 if out.shape[-1] != skip.shape[-1] else 0
        pad_b = 1 if out.shape[-2] != skip.shape[-2] else 0
        if pad_r or pad_b:
            out = F.pad(out, [0, pad_r, 0, pad_b], "reflect")
        out = torch.cat([out, skip], dim=1)
        if i < num_pools - 1:
            out = conv_block(out, p[f"{pre}up_conv.{i}.layers.0.weight"],
                             p[f"{pre}up_conv.{i}.layers.3.weight"])
        else:
            out = conv_block(out, p[f"{pre}up_conv.{i}.0.layers.0.weight"],
                             p[f"{pre}up_conv.{i}.0.layers.3.weight"])
            out = F.conv2d(out, p[f"{pre}up_conv.{i}.1.weight"], p[f"{pre}up_conv.{i}.1.bias"])
    return out


def group_norm_stats(x2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-sample mean / UNBIASED std over the {real, imag} channel groups.
    Reference: varnet.py:257-268.  x2 is [B, 2c, H, W]."""
    b, c, h, w = x2.shape
    g = x2.reshape(b, 2, (c // 2) * h * w)
    mean = g.mean(dim=2).view(b, 2, 1, 1)
    std = g.std(dim=2).view(b, 2, 1, 1)
    return mean, std


def _pad16(x: torch.Tensor):
    """Zero pad H, W up to the next multiple of 16 (centred, floor left).
    Reference: varnet.py:275-289."""
    _, _, h, w = x.shape
    wm = ((w - 1) | 15) + 1
    hm = ((h - 1) | 15) + 1
    wp = [(wm - w) // 2, (wm - w) - (wm - w) // 2]
    hp = [(hm - h) // 2, (hm - h) - (hm - h) // 2]
    return F.pad(x, wp + hp), (hp, wp, hm, wm)


def normunet_forward(p: Params, pre: str, x: torch.Tensor, ref: Optional[torch.Tensor],
                     num_pools: int, use_ref: bool) -> torch.Tensor:
    """NormUnet on a complex [B, c, H, W] tensor.  Reference: varnet.py:301-332."""
    assert torch.is_complex(x)
    c = x.shape[1]
    x2 = torch.cat([x.real, x.imag], dim=1)
    mean, std = group_norm_stats(x2)
    # mean/std are [B,2,1,1]; with c == 1 they broadcast over the 2 channels,
    # for c > 1 the reference's broadcasting requires c == 1 (it asserts nothing
    # but (B,2c,H,W) - (B,2,1,1) only broadcasts when 2c == 2): keep that.
    x2 = (x2 - mean) / (std + 1e-6)
    x2, (hp, wp, hm, wm) = _pad16(x2)
    if use_ref:
        assert ref is not None and not torch.is_complex(ref)
        r = _inorm(ref)
        r, _ = _pad16(r)
        x2 = torch.cat([x2, r], dim=1)
    else:
        assert ref is None
    y = unet_forward(p, pre + "unet.", x2, num_pools)
    y = y[..., hp[0]:hm - hp[1], wp[0]:wm - wp[1]]
    y = y * std + mean
    return torch.complex(y[:, :c].contiguous(), y[:, c:].contiguous())


# --------------------------------------------------------------------------
# VarNet
# --------------------------------------------------------------------------


def acs_mask(width: int, num_low_frequencies: int, dtype=torch.float32) -> torch.Tensor:
    """1-D low-frequency window: ones on [0, nlf) rolled by (-nlf)//2 (Python
    floor division of the negated value).  Reference: varnet.py:395-397."""
    m = torch.ones(width, dtype=dtype)
    m[num_low_frequencies:] = 0
    return torch.roll(m, (-num_low_frequencies) // 2)


def sensitivity_forward(p: Params, pre: str, masked_kspace: torch.Tensor,
                        num_low_frequencies: int, num_pools: int) -> torch.Tensor:
    """Coil sensitivity estimate.  Reference: varnet.py:389-420."""
    rdtype = masked_kspace.real.dtype
    m = acs_mask(masked_kspace.shape[-1], num_low_frequencies, rdtype)
    acs = ifft2(masked_kspace * m[None, None, None, :])
    n, c, h, w = acs.shape
    planes = acs.reshape(n * c, 1, h, w)
    est = normunet_forward(p, pre + "norm_unet.", planes, None, num_pools, False)
    est = est.reshape(n, c, h, w)
    return est / (rss(est) + 1e-6)


def sens_reduce(kspace: torch.Tensor, sens: torch.Tensor) -> torch.Tensor:
    """Reference: varnet.py:511-512."""
    return (ifft2(kspace) * sens.conj()).sum(dim=1, keepdim=True)


def sens_expand(image: torch.Tensor, sens: torch.Tensor) -> torch.Tensor:
    """Reference: varnet.py:508-509."""
    return fft2(image * sens)


def varnet_block_forward(p: Params, pre: str, k: torch.Tensor, k0: torch.Tensor,
                         mask: torch.Tensor, sens: torch.Tensor, ref: Optional[torch.Tensor],
                         num_pools: int, use_ref: bool) -> torch.Tensor:
    """One cascade: k - w*where(M, k-k0, 0) - expand(NormUnet(reduce(k), ref)).
    Reference: varnet.py:514-530.  ``mask`` is bool, broadcastable to k."""
    m = sens_reduce(k, sens)
    m = normunet_forward(p, pre + "model.", m, ref if use_ref else None, num_pools, use_ref)
    model_term = sens_expand(m, sens)
    zero = torch.zeros(1, 1, 1, 1, dtype=k.dtype)
    soft_dc = torch.where(mask, k - k0, zero) * p[pre + "dc_weight"]
    return k - soft_dc - model_term


def varnet_forward(p: Params, masked_kspace: torch.Tensor, mask: torch.Tensor,
                   ref: Optional[torch.Tensor], num_low_frequencies: int, *,
                   num_cascades: int, pools: int = 4, sens_pools: int = 4,
                   use_ref: bool = True, pre: str = "",
                   return_intermediates: bool = False):
    """Full VarNet.  Reference: varnet.py:465-486."""
    sens = sensitivity_forward(p, pre + "sens_net.", masked_kspace, num_low_frequencies, sens_pools)
    k = masked_kspace.clone()
    if use_ref:
        ref = rss(ref)
    inter = [k]
    for j in range(num_cascades):
        k = varnet_block_forward(p, f"{pre}cascades.{j}.", k, masked_kspace, mask, sens,
                                 ref, pools, use_ref)
        inter.append(k)
    out = rss(ifft2(k))
    if return_intermediates:
        return out, sens, inter
    return out


# --------------------------------------------------------------------------
# alignment network (BatchNorm U-Net) + warp
# --------------------------------------------------------------------------


class BNState:
    """Collects BatchNorm batch statistics seen during a training-mode pass so a
    test can compare running-stat updates.  Reference semantics: unet.py:125,
    torch BatchNorm2d(eps=1e-5, momentum=0.1)."""

    def __init__(self):
        self.batch_mean: Dict[str, torch.Tensor] = {}
        self.batch_var_unbiased: Dict[str, torch.Tensor] = {}


def _bn(p: Params, pre: str, x: torch.Tensor, training: bool, state: Optional[BNState]) -> torch.Tensor:
    w, b = p[pre + "weight"], p[pre + "bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if state is not None:
            n = x.numel() // x.shape[1]
            state.batch_mean[pre] = mean
            state.batch_var_unbiased[pre] = var * (n / max(n - 1, 1))
    else:
        mean, var = p[pre + "running_mean"], p[pre + "running_var"]
    scale = w / torch.sqrt(var + 1e-5)
    return (x - mean[None, :, None, None]) * scale[None, :, None, None] + b[None, :, None, None]


def _cba(p: Params, pre: str, x: torch.Tensor, training: bool, state, conv_i: int = 0) -> torch.Tensor:
    """conv(+bias) -> BatchNorm -> LeakyReLU(0.01).  conv_i is the index of the
    conv inside the Sequential (0 for Conv2d(), 1 for Up()/Down()).
    Reference: unet.py:119-140."""
    w = p[f"{pre}{conv_i}.weight"]
    x = F.conv2d(x, w, p[f"{pre}{conv_i}.bias"], padding=w.shape[-1] // 2)
    x = _bn(p, f"{pre}{conv_i + 1}.", x, training, state)
    return F.leaky_relu(x, 0.01)


def _res(p: Params, pre: str, x: torch.Tensor, training: bool, state) -> torch.Tensor:
    """x + chain of conv-bn-act under ``pre``subnet.  Reference: unet.py:15-24."""
    y = x
    i = 0
    while f"{pre}subnet.{i}.0.weight" in p:
        y = _cba(p, f"{pre}subnet.{i}.", y, training, state)
        i += 1
    return x + y


def _align_level(p: Params, pre: str, x: torch.Tensor, training: bool, state) -> torch.Tensor:
    """One CatSequential level: cat([module(x), x]).  Reference: unet.py:6-13,149-170."""
    m = pre + "module."
    y = F.avg_pool2d(x, 2, 2)
    y = _cba(p, m + "0.", y, training, state, conv_i=1)           # Down
    y = _res(p, m + "1.", y, training, state)
    if f"{m}2.module.0.1.weight" in p:                              # has an inner level
        y = _align_level(p, m + "2.", y, training, state)
        y = _cba(p, m + "3.", y, training, state)
        y = _res(p, m + "4.", y, training, state)
        up = m + "5."
    else:
        up = m + "2."
    y = F.interpolate(y, scale_factor=2, mode="nearest")            # Up
    y = _cba(p, up, y, training, state, conv_i=1)
    return torch.cat([y, x], dim=1)


def align_unet_forward(p: Params, pre: str, x: torch.Tensor, training: bool = False,
                       state: Optional[BNState] = None) -> torch.Tensor:
    """unet.UNet.forward.  Reference: unet.py:145-189.  ``pre`` ends in 'unet.'"""
    y = _cba(p, pre + "0.", x, training, state)
    y = _res(p, pre + "1.", y, training, state)
    y = _align_level(p, pre + "2.", y, training, state)
    y = _cba(p, pre + "3.", y, training, state)
    y = _res(p, pre + "4.", y, training, state)
    return F.conv2d(y, p[pre + "5.weight"], p[pre + "5.bias"], padding=1)


def identity_grid(h: int, w: int, dtype=torch.float32) -> torch.Tensor:
    """affine_grid(identity, align_corners=False): x_j=(2j+1)/W-1, y_i=(2i+1)/H-1,
    last axis (x, y).  Reference: cross.py:24-26."""
    xs = (torch.arange(w, dtype=dtype) * 2 + 1) / w - 1
    ys = (torch.arange(h, dtype=dtype) * 2 + 1) / h - 1
    g = torch.empty(1, h, w, 2, dtype=dtype)
    g[0, :, :, 0] = xs[None, :]
    g[0, :, :, 1] = ys[:, None]
    return g


def spatial_transformer_forward(p: Params, moving: torch.Tensor, fixed: torch.Tensor,
                                training: bool = False, state: Optional[BNState] = None,
                                pre: str = "") -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (offset [N,H,W,2], grid [N,H,W,2]).  Reference: cross.py:23-30."""
    x = torch.cat([moving, fixed], dim=1)
    y = align_unet_forward(p, pre + "net.0.unet.", x, training, state)
    y = F.leaky_relu(y, 0.01)
    y = F.conv2d(y, p[pre + "net.2.weight"], p[pre + "net.2.bias"], padding=1)
    offset = y.permute(0, 2, 3, 1)
    grid = identity_grid(moving.shape[2], moving.shape[3], moving.dtype) + offset
    return offset, grid


def warp(img: torch.Tensor, grid: torch.Tensor, padding_mode: str = "zeros") -> torch.Tensor:
    """Bilinear grid_sample, zeros padding, align_corners=False.
    Reference: cross.py:32-34 (augment.py:60-61 uses 'reflection')."""
    return F.grid_sample(img, grid, mode="bilinear", padding_mode=padding_mode, align_corners=False)


def warp_manual(img: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """Loop-free restatement of the bilinear/zeros sampler from first principles
    (ix=((x+1)W-1)/2), used to pin ``warp`` and as the spec for the HIP kernel."""
    n, c, h, w = img.shape
    ix = ((grid[..., 0] + 1) * w - 1) / 2
    iy = ((grid[..., 1] + 1) * h - 1) / 2
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    out = torch.zeros(n, c, grid.shape[1], grid.shape[2], dtype=img.dtype)
    flat = img.reshape(n, c, h * w)
    for dy in (0, 1):
        for dx in (0, 1):
            xx = x0 + dx
            yy = y0 + dy
            wgt = (1 - (ix - xx).abs()) * (1 - (iy - yy).abs())
            ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
            idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).long().reshape(n, 1, -1).expand(n, c, -1)
            v = torch.gather(flat, 2, idx).reshape(n, c, grid.shape[1], grid.shape[2])
            out = out + v * (wgt * ok.to(img.dtype))[:, None]
    return out


# --------------------------------------------------------------------------
# augmentation (augment.py:7-66) -- random draws are ARGUMENTS here
# --------------------------------------------------------------------------


def rigid_affine(r_s, t_s, dtype=torch.float32) -> torch.Tensor:
    """[N,2,3] matrices M = T @ R (rotation r about the centre, then translation t on both axes).
    Reference: augment.py:7-33 (built in float64 numpy, cast to the image dtype)."""
    import numpy as np
    mats = []
    for r, t in zip(r_s, t_s):
        rot = np.array([[np.cos(r), -np.sin(r), 0.0], [np.sin(r), np.cos(r), 0.0], [0.0, 0.0, 1.0]])
        tr = np.array([[1.0, 0.0, t], [0.0, 1.0, t], [0.0, 0.0, 1.0]])
        mats.append((tr @ rot)[:-1])
    return torch.as_tensor(np.stack(mats, 0), dtype=dtype)


def rigid_grid(m: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """affine_grid(M, align_corners=False): grid[n,i,j] = M_n @ (x_j, y_i, 1).  augment.py:35-38."""
    base = identity_grid(h, w, m.dtype)[0]                               # [h, w, 2]
    ones = torch.ones(h, w, 1, dtype=m.dtype)
    hom = torch.cat([base, ones], dim=-1)                                # [h, w, 3]
    return torch.einsum("nij,hwj->nhwi", m, hom)


def _cubic_weights(t: torch.Tensor, a: float = -0.75):
    """Cubic convolution coefficients for taps at -1, 0, +1, +2 (Keys, A = -0.75: ATen's bicubic)."""
    def c1(x):   # |x| <= 1
        return ((a + 2) * x - (a + 3)) * x * x + 1
    def c2(x):   # 1 < |x| < 2
        return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    return [c2(t + 1), c1(t), c1(1 - t), c2(2 - t)]


def bicubic_upsample(x: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """F.interpolate(x, size=(h, w), mode='bicubic', align_corners=False) from first principles:
    src = (dst + 0.5) * in/out - 0.5 (not clamped), taps floor(src)-1 .. +2 clamped to the border."""
    n, c, hi, wi = x.shape
    def axis(out_len, in_len):
        src = (torch.arange(out_len, dtype=x.dtype) + 0.5) * (in_len / out_len) - 0.5
        i0 = torch.floor(src)
        wts = _cubic_weights(src - i0)
        idx = [(i0 + k - 1).clamp(0, in_len - 1).long() for k in range(4)]
        return idx, wts
    iy, wy = axis(h, hi)
    ix, wx = axis(w, wi)
    out = torch.zeros(n, c, h, w, dtype=x.dtype)
    for a_ in range(4):
        rows = x[:, :, iy[a_], :]                                       # [n, c, h, wi]
        for b_ in range(4):
            out = out + rows[:, :, :, ix[b_]] * (wy[a_][:, None] * wx[b_][None, :])
    return out


def bspline_grid(ctrl: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """ctrl [N,2,9,9] = (rand - 0.5) * 2 / 50 -> NHWC offsets [N,h,w,2].  augment.py:40-48."""
    return bicubic_upsample(ctrl, h, w).permute(0, 2, 3, 1).contiguous()


def sample_reflect(img: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """grid_sample(bilinear, padding_mode='reflection', align_corners=False), real input.  augment.py:60-61.
    Reflection is about the pixel EDGES (-0.5, size-0.5), then clipped to [0, size-1]."""
    n, c, h, w = img.shape
    def unnorm_reflect(g, size):
        x = ((g + 1) * size - 1) / 2
        lo, span = -0.5, float(size)
        x = (x - lo).abs()
        flips = torch.floor(x / span)
        extra = x - flips * span
        x = torch.where(flips.long() % 2 == 0, extra + lo, span - extra + lo)
        return x.clamp(0, size - 1)
    ix = unnorm_reflect(grid[..., 0], w)
    iy = unnorm_reflect(grid[..., 1], h)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    out = torch.zeros(n, c, grid.shape[1], grid.shape[2], dtype=img.dtype)
    flat = img.reshape(n, c, h * w)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            wgt = (1 - (ix - xx).abs()) * (1 - (iy - yy).abs())
            ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
            idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).long().reshape(n, 1, -1).expand(n, c, -1)
            v = torch.gather(flat, 2, idx).reshape(n, c, grid.shape[1], grid.shape[2])
            out = out + v * (wgt * ok.to(img.dtype))[:, None]
    return out


def augment(img: torch.Tensor, r_s=None, t_s=None, ctrl: Optional[torch.Tensor] = None,
            grid: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """augment.py:50-66 with the random draws passed in: rigid (r_s, t_s) [+ B-spline ctrl], or a given grid."""
    real_dtype = img.real.dtype if torch.is_complex(img) else img.dtype
    if grid is None:
        grid = rigid_grid(rigid_affine(r_s, t_s, real_dtype), img.shape[2], img.shape[3])
        if ctrl is not None:
            grid = grid + bspline_grid(ctrl.to(real_dtype), img.shape[2], img.shape[3])
    if torch.is_complex(img):
        out = torch.complex(sample_reflect(img.real.contiguous(), grid), sample_reflect(img.imag.contiguous(), grid))
    else:
        out = sample_reflect(img, grid)
    return out, grid


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------


def ssimloss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """1 - mean SSIM, 7x7 uniform valid window, data_range 1, cov_norm 49/48.
    Reference: ssimloss.py:11-40."""
    win = 7
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    npx = win * win
    cov = npx / (npx - 1)
    k = torch.ones(1, 1, win, win, dtype=x.dtype) / npx
    ux, uy = F.conv2d(x, k), F.conv2d(y, k)
    uxx, uyy, uxy = F.conv2d(x * x, k), F.conv2d(y * y, k), F.conv2d(x * y, k)
    vx, vy, vxy = cov * (uxx - ux * ux), cov * (uyy - uy * uy), cov * (uxy - ux * uy)
    a1, a2 = 2 * ux * uy + c1, 2 * vxy + c2
    b1, b2 = ux * ux + uy * uy + c1, vx + vy + c2
    return 1 - ((a1 * a2) / (b1 * b2)).mean()


def lncc_loss(i: torch.Tensor, j: torch.Tensor, win: int = 9) -> torch.Tensor:
    """-mean(cross^2 / (Ivar*Jvar + 1e-5)) with win x win zero-padded box sums.
    Reference: lnccloss.py:7-56."""
    k = torch.ones(1, 1, win, win, dtype=i.dtype)
    pad = win // 2
    s = lambda t: F.conv2d(t, k, padding=pad)
    i_s, j_s, ii_s, jj_s, ij_s = s(i), s(j), s(i * i), s(j * j), s(i * j)
    n = float(win * win)
    ui, uj = i_s / n, j_s / n
    cross = ij_s - uj * i_s - ui * j_s + ui * uj * n
    ivar = ii_s - 2 * ui * i_s + ui * ui * n
    jvar = jj_s - 2 * uj * j_s + uj * uj * n
    return -(cross * cross / (ivar * jvar + 1e-5)).mean()


def gaussian_kernel_2d(sigma: float, dtype=torch.float32) -> torch.Tensor:
    """Outer product of two normalised 1-D Gaussians, 2*ceil(2 sigma)+1 taps.
    Reference: miloss.py:6-18."""
    size = int(2 * math.ceil(sigma * 2) + 1)
    x = torch.linspace(-(size - 1) // 2, (size - 1) // 2, size)
    g = 1.0 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    k = torch.outer(g, g)
    return (k / k.sum()).to(dtype)


def ms_lncc_loss(i: torch.Tensor, j: torch.Tensor, win: int = 9, ms: int = 3, sigma: float = 3) -> torch.Tensor:
    """Multi-scale LNCC: Gaussian smooth + 2x average pool between scales.
    Reference: lnccloss.py:58-65, miloss.py:20-24."""
    g = gaussian_kernel_2d(sigma, i.dtype)[None, None]
    down = lambda t: F.avg_pool2d(F.conv2d(t, g, padding=g.shape[-1] // 2), 2, 2)
    loss = lncc_loss(i, j, win)
    for _ in range(ms - 1):
        i, j = down(i), down(j)
        loss = loss + lncc_loss(i, j, win)
    return loss / ms


def gradient_loss(s: torch.Tensor) -> torch.Tensor:
    """(mean(dW^2) + mean(dH^2)) / 2 of an NHWC offset field.
    Reference: model.py:21-28."""
    assert s.shape[-1] == 2
    dx = s[:, :, 1:, :] - s[:, :, :-1, :]
    dy = s[:, 1:, :, :] - s[:, :-1, :, :]
    return ((dx * dx).mean() + (dy * dy).mean()) / 2.0


# --------------------------------------------------------------------------
# model-level glue (CSModel.set_input / forwardT / forwardR)
# --------------------------------------------------------------------------


def set_input(img_full: torch.Tensor, img_aux: torch.Tensor, pruned: torch.Tensor) -> Dict[str, torch.Tensor]:
    """fft2 -> drop pruned columns -> ifft2 -> rss.  Reference: model.py:108-121."""
    keep = (~pruned).to(img_full.real.dtype)
    k_full = fft2(img_full)
    k_samp = k_full * keep
    samp = ifft2(k_samp)
    return {
        "img_k_full": k_full, "img_k_sampled": k_samp, "img_sampled": samp,
        "img_full_rss": rss(img_full), "img_sampled_rss": rss(samp), "img_aux_rss": rss(img_aux),
        "img_mask": fftshift2(torch.ones_like(rss(img_full)) - pruned.to(img_full.real.dtype)),
    }


def recon_align_forward(p_T: Params, p_R: Params, img_full: torch.Tensor, img_aux: torch.Tensor,
                        pruned: torch.Tensor, *, shape: int, sparsity: float, num_cascades: int,
                        weight_smooth: float = 1000.0, weight_sim: float = 1.0,
                        pools: int = 4, sens_pools: int = 4,
                        training: bool = False, state: Optional[BNState] = None) -> Dict[str, torch.Tensor]:
    """set_input + forwardT + forwardR ('Rec' regime loss).  Reference:
    model.py:89-121 (set_input), :142-155 (forwardT), :157-169 (forwardR)."""
    o = set_input(img_full, img_aux, pruned)
    offset, grid = spatial_transformer_forward(p_T, img_aux.abs(), o["img_sampled"].abs(), training, state)
    warped = warp(img_aux.abs(), grid)
    o["img_offset"], o["img_grid"], o["img_warped"] = offset, grid, warped
    o["img_warped_rss"] = rss(warped)
    o["loss_smooth"] = gradient_loss(offset)
    nlf = int(shape * sparsity * 0.32)
    rec = varnet_forward(p_R, o["img_k_sampled"], torch.logical_not(pruned), warped, nlf,
                         num_cascades=num_cascades, pools=pools, sens_pools=sens_pools, use_ref=True)
    o["img_rec"] = rec
    o["loss_sim"] = ssimloss(o["img_full_rss"], rec)
    o["loss_all"] = o["loss_smooth"] * weight_smooth + o["loss_sim"] * weight_sim
    return o


# --------------------------------------------------------------------------
# reporting metrics (skimage is absent in this image; numpy restatements)
# --------------------------------------------------------------------------


def metric_mse(gt: torch.Tensor, pred: torch.Tensor) -> float:
    """metrics.py:23-25."""
    return ((gt.double() - pred.double()) ** 2).mean().item()


def metric_mae(gt: torch.Tensor, pred: torch.Tensor) -> float:
    """metrics.py:27-29."""
    return (gt.double() - pred.double()).abs().mean().item()


def metric_nmse(gt: torch.Tensor, pred: torch.Tensor) -> float:
    """metrics.py:31-33."""
    return (((gt.double() - pred.double()) ** 2).sum() / (gt.double() ** 2).sum()).item()


def metric_mi(gt: torch.Tensor, pred: torch.Tensor, bins: int = 64) -> float:
    """Mutual information from a bins x bins joint histogram over [0, 1]^2 per image, batch mean
    (metrics.py:55-69; np.histogram2d: samples outside the range are dropped, 1.0 goes to the last bin)."""
    import numpy as np
    vals = []
    for x, y in zip(gt.double().numpy(), pred.double().numpy()):
        x, y = x.ravel(), y.ravel()
        keep = (x >= 0) & (x <= 1) & (y >= 0) & (y <= 1)
        bx = np.minimum(np.floor(x[keep] * bins).astype(np.int64), bins - 1)
        by = np.minimum(np.floor(y[keep] * bins).astype(np.int64), bins - 1)
        pxy = np.bincount(bx * bins + by, minlength=bins * bins).reshape(bins, bins).astype(np.float64)
        pxy = pxy / (pxy.sum() + 1e-10)
        px, py = pxy.sum(axis=1), pxy.sum(axis=0)
        pp = px[:, None] * py[None, :]
        nz = pxy > 0
        vals.append(float((pxy[nz] * np.log(pxy[nz])).sum() - (pxy[nz] * np.log(pp[nz])).sum()))
    return float(np.mean(vals))


def psnr(gt: torch.Tensor, pred: torch.Tensor, data_range: float = 1.0) -> float:
    """One PSNR over the whole batch.  Reference: metrics.py:35-38."""
    err = ((gt.double() - pred.double()) ** 2).mean().item()
    return 10.0 * math.log10(data_range ** 2 / err) if err > 0 else float("inf")


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    d = (a.double() - b.double())
    return (d.abs().pow(2).sum().sqrt() / b.double().abs().pow(2).sum().sqrt()).item()
