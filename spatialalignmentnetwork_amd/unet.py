"""Host-side mirror of the reference's alignment backbone (unet.py:6-24,119-189).

Same factory names (``Conv2d``, ``Up``, ``Down``), container classes
(``CatSequential.module``, ``ResSequential.subnet``) and therefore the same
state_dict keys as the reference, but executed by the HIP kernels with
BatchNorm + LeakyReLU(0.01) applied lazily by whichever kernel reads the tensor:

  * conv (+bias) writes RAW output; BatchNorm becomes a per-channel affine
    (eval: from running stats; train: from the conv epilogue's tile statistics);
  * ``cat([module(x), x])`` is zero-copy: both producers write at channel
    offsets of one buffer;
  * ``Up`` = nearest x2 -> conv1x1 -> BN -> act is evaluated as conv1x1 at LOW
    resolution followed by the up-sampling materialiser (a 1x1 conv commutes
    with nearest up-sampling and the batch statistics are identical), 4x fewer
    MACs than the reference's order.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from .ops import Act, GLOBAL_ARENA as ARENA

BN_EPS = 1e-5
SLOPE = 0.01   # nn.LeakyReLU default, unet.py:126


class CatSequential(torch.nn.Module):
    """cat([module(x), x], dim).  Reference: unet.py:6-13."""

    def __init__(self, *modules, dim=1):
        super().__init__()
        self.module = torch.nn.Sequential(*modules)
        self.dim = dim

    def forward(self, x):
        raise RuntimeError("CatSequential is executed by UNet.run (fused HIP path); it has no stand-alone forward")


class ResSequential(torch.nn.Module):
    """x + subnet(x).  Reference: unet.py:15-24."""

    def __init__(self, *modules, sample=None):
        super().__init__()
        self.subnet = torch.nn.Sequential(*modules)
        self.sample = sample

    def forward(self, x):
        raise RuntimeError("ResSequential is executed by UNet.run (fused HIP path); it has no stand-alone forward")


def Conv2d(in_channels, out_channels):
    """conv3x3(bias) + BatchNorm2d + LeakyReLU.  Reference: unet.py:119-126."""
    return torch.nn.Sequential(
        torch.nn.Conv2d(in_channels, out_channels, 3, padding=1),
        torch.nn.BatchNorm2d(out_channels),
        torch.nn.LeakyReLU(inplace=True))


def Up(in_channels, out_channels):
    """nearest x2 + conv1x1 + BN + LeakyReLU.  Reference: unet.py:128-133."""
    return torch.nn.Sequential(
        torch.nn.Upsample(scale_factor=(2, 2)),
        torch.nn.Conv2d(in_channels, out_channels, kernel_size=1),
        torch.nn.BatchNorm2d(out_channels),
        torch.nn.LeakyReLU(inplace=True))


def Down(in_channels, out_channels):
    """avgpool2 + conv1x1 + BN + LeakyReLU.  Reference: unet.py:135-140."""
    return torch.nn.Sequential(
        torch.nn.AvgPool2d(2, stride=2),
        torch.nn.Conv2d(in_channels, out_channels, kernel_size=1),
        torch.nn.BatchNorm2d(out_channels),
        torch.nn.LeakyReLU(inplace=True))


class UNet(torch.nn.Module):
    """Recursive Cat/Res U-Net.  Reference: unet.py:144-189."""

    def __init__(self, in_channels, out_channels, layers):
        super().__init__()
        layers = list(layers)
        self.layer_channels = list(layers)
        num_convs = 2
        current_layer = layers.pop()
        upper_layer = layers.pop()
        unet = CatSequential(
            Down(upper_layer, current_layer),
            ResSequential(*[Conv2d(current_layer, current_layer) for _ in range(num_convs)]),
            Up(current_layer, current_layer))
        for layer in reversed(layers):
            lower_layer, current_layer, upper_layer = current_layer, upper_layer, layer
            unet = CatSequential(
                Down(upper_layer, current_layer),
                ResSequential(*[Conv2d(current_layer, current_layer) for _ in range(num_convs)]),
                unet,
                Conv2d(current_layer + lower_layer, current_layer),
                ResSequential(*[Conv2d(current_layer, current_layer) for _ in range(num_convs - 1)]),
                Up(current_layer, current_layer))
        lower_layer, current_layer = current_layer, upper_layer
        self.unet = torch.nn.Sequential(
            Conv2d(in_channels, current_layer),
            ResSequential(*[Conv2d(current_layer, current_layer) for _ in range(num_convs - 1)]),
            unet,
            Conv2d(current_layer + lower_layer, current_layer),
            ResSequential(*[Conv2d(current_layer, current_layer) for _ in range(num_convs - 1)]),
            torch.nn.Conv2d(current_layer, out_channels, 3, padding=1))
        self.in_channels, self.out_channels = in_channels, out_channels

    # ------------------------------------------------------------------ fused executor
    def _cba(self, seq, conv_i: int, x: Act, out: Act, tag: str, count_scale: int = 1) -> Act:
        """conv(+bias) raw into ``out`` and the lazy BatchNorm affine into out.scale/shift."""
        conv, bn = seq[conv_i], seq[conv_i + 1]
        if bn.training:
            part = ops.conv2d(x, conv.weight, conv.bias, out, stats=True, tag=tag)
            c = conv.weight.shape[0]
            bmean = ARENA.get(f"{tag}.bmean", (c,), x.buf.device)
            bvar = ARENA.get(f"{tag}.bvar", (c,), x.buf.device)
            ops.norm_finalize(part, ops.NORM_BATCH, BN_EPS, out.scale, out.shift, out.coff, gamma=bn.weight.detach(),
                              beta=bn.bias.detach(), aux_a=bmean, aux_b=bvar)
            _update_running_stats(bn, bmean, bvar, x.n * x.h * x.w, count_scale)
        else:
            ops.conv2d(x, conv.weight, conv.bias, out, stats=False)
            ops.bn_eval_affine(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, BN_EPS,
                               out.scale, out.shift, out.coff)
        return out

    def _res(self, res: ResSequential, x: Act, out: Act, key: str) -> Act:
        """out = x + subnet(x); x is read lazily, out is materialised (identity affine)."""
        cur = x
        for i, seq in enumerate(res.subnet):
            t = _arena_act(f"{key}.r{i}", x.n, seq[0].weight.shape[0], x.h, x.w, x.buf.device)
            cur = self._cba(seq, 0, cur, t, key)
        ops.add(x, cur, out)
        return out

    def _level(self, cat: CatSequential, x: Act, up_out: Act, key: str) -> None:
        """Run ``cat.module`` on x and write its (up-sampled, activated) result into up_out."""
        m = cat.module
        n, h, w, dev = x.n, x.h // 2, x.w // 2, x.buf.device
        c_cur = m[0][1].weight.shape[0]
        pooled = Act(ARENA.get(f"{key}.pool", (n, x.c, h, w), dev), 0, x.c)
        ops.avgpool2(x, pooled)
        d = _arena_act(f"{key}.down", n, c_cur, h, w, dev)
        self._cba(m[0], 1, pooled, d, key)
        has_inner = isinstance(m[2], CatSequential)
        if has_inner:
            c_low = m[2].module[-1][1].weight.shape[0]
            z = Act(ARENA.get(f"{key}.cat", (n, c_low + c_cur, h, w), dev), 0, c_low + c_cur)
            r = self._res(m[1], d, z.view(c_low, c_cur), key + ".res1")
            self._level(m[2], r, z.view(0, c_low), key + "d")
            t = _arena_act(f"{key}.merge", n, c_cur, h, w, dev)
            self._cba(m[3], 0, z, t, key)
            r2 = Act(ARENA.get(f"{key}.res2", (n, c_cur, h, w), dev), 0, c_cur)
            self._res(m[4], t, r2, key + ".res2")
            up_seq, src = m[5], r2
        else:
            r = Act(ARENA.get(f"{key}.res1o", (n, c_cur, h, w), dev), 0, c_cur)
            self._res(m[1], d, r, key + ".res1")
            up_seq, src = m[2], r
        # Up: conv1x1 at low resolution, then nearest x2 with BN + act applied on the fly
        u = _arena_act(f"{key}.uplow", n, up_seq[1].weight.shape[0], h, w, dev)
        self._cba(up_seq, 1, src, u, key, count_scale=4)
        ops.upsample2(u, up_out)

    def run(self, x: Act, out: Act, key: str = "align") -> Act:
        """x: materialised [N, in_channels, H, W]; out: raw [N, out_channels, H, W]."""
        s = self.unet
        n, h, w, dev = x.n, x.h, x.w, x.buf.device
        depth = len(self.layer_channels) - 1
        if (h % (1 << depth)) or (w % (1 << depth)):
            raise NotImplementedError(f"alignment U-Net needs H, W divisible by {1 << depth}, got {h}x{w}")
        c0 = s[0][0].weight.shape[0]
        c_low = s[2].module[-1][1].weight.shape[0]
        t0 = _arena_act(f"{key}.t0", n, c0, h, w, dev)
        self._cba(s[0], 0, x, t0, key)
        z = Act(ARENA.get(f"{key}.cat0", (n, c_low + c0, h, w), dev), 0, c_low + c0)
        r = self._res(s[1], t0, z.view(c_low, c0), key + ".res1")
        self._level(s[2], r, z.view(0, c_low), key + "d")
        t = _arena_act(f"{key}.merge0", n, c0, h, w, dev)
        self._cba(s[3], 0, z, t, key)
        r2 = Act(ARENA.get(f"{key}.res2_0", (n, c0, h, w), dev), 0, c0)
        self._res(s[4], t, r2, key + ".res2")
        ops.conv2d(r2, s[5].weight, s[5].bias, out, stats=False)
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, _, h, w = x.shape
        y = torch.empty((n, self.out_channels, h, w), device=x.device)
        self.run(ops.full(x.contiguous()), ops.full(y))
        return y


def _arena_act(name, n, c, h, w, dev) -> Act:
    return Act(ARENA.get(name, (n, c, h, w), dev), 0, c, ARENA.get(name + ".sc", (n, c), dev),
               ARENA.get(name + ".sh", (n, c), dev), SLOPE)


def _update_running_stats(bn: torch.nn.BatchNorm2d, bmean: torch.Tensor, bvar_unbiased: torch.Tensor, count: int,
                          count_scale: int) -> None:
    """running = (1-m)*running + m*batch (m = 0.1), unbiased batch variance.
    ``count_scale`` = 4 for the Up blocks, whose statistics are taken at low
    resolution: the reference sees every value 4 times, which changes only the
    n/(n-1) factor of the unbiased variance."""
    with torch.no_grad():
        m = bn.momentum if bn.momentum is not None else 0.1
        if count_scale != 1 and count > 1:
            big = count * count_scale
            bvar_unbiased = bvar_unbiased * ((count - 1) / count) * (big / (big - 1))
        bn.running_mean.mul_(1 - m).add_(bmean, alpha=m)
        bn.running_var.mul_(1 - m).add_(bvar_unbiased, alpha=m)
        bn.num_batches_tracked += 1
