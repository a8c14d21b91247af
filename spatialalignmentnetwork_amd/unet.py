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

from typing import List, Optional  # noqa: F401

import torch

from . import ops
from .ops import Act, GLOBAL_ARENA as ARENA

# A residual add's gradient goes to both operands as ONE tensor (no copy per reader; round 6).  SAN_UNET_SHARE_GRADS=0: a copy per reader.
SHARE_RES_GRADS = [__import__("os").environ.get("SAN_UNET_SHARE_GRADS", "1") != "0"]
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
        self._tape = None
        self._produced = {}

    # ------------------------------------------------------------------ fused executor
    # Every op goes through one of five recorders (conv+BN, bare conv, add, avg-pool, up-sample) so
    # that run_bwd() can replay the tape in reverse; activations are identified by (buffer, channel
    # range), gradients are materialised per activation and summed when an activation has two readers.
    def _cba(self, seq, conv_i: int, x: Act, out: Act, tag: str, count_scale: int = 1) -> Act:
        """conv(+bias) raw into ``out`` and the lazy BatchNorm affine into out.scale/shift."""
        conv, bn = seq[conv_i], seq[conv_i + 1]
        if bn.training:
            c = conv.weight.shape[0]
            bmean = ARENA.get(f"{tag}.bmean", (c,), x.buf.device)
            bvar = ARENA.get(f"{tag}.bvar", (c,), x.buf.device)
            m, factor = _running_factors(bn, x.n * x.h * x.w, count_scale)
            part = ops.conv2d(x, conv.weight, conv.bias, out, stats=True, tag=tag)
            ops.norm_finalize_bn(part, BN_EPS, out.scale, out.shift, out.coff, bn, bmean, bvar, m, factor)
        else:
            ops.conv2d(x, conv.weight, conv.bias, out, stats=False)
            ops.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, BN_EPS,
                               out.scale, out.shift, out.coff)
        self._record(("cba", conv, bn, x, out), out)
        return out

    def _record(self, op, out: Act) -> None:
        if self._tape is not None:
            self._tape.append(op)
            self._produced.setdefault(out.buf.data_ptr(), []).append((out.coff, out.c))

    def _add(self, a: Act, b: Act, out: Act) -> None:
        ops.add(a, b, out)
        self._record(("add", a, b, out), out)

    def _res(self, res: ResSequential, x: Act, out: Act, key: str) -> Act:
        """out = x + subnet(x); x is read lazily, out is materialised (identity affine)."""
        cur = x
        for i, seq in enumerate(res.subnet):
            t = _arena_act(f"{key}.r{i}", x.n, seq[0].weight.shape[0], x.h, x.w, x.buf.device)
            cur = self._cba(seq, 0, cur, t, key)
        self._add(x, cur, out)
        return out

    def _level(self, cat: CatSequential, x: Act, up_out: Act, key: str) -> None:
        """Run ``cat.module`` on x and write its (up-sampled, activated) result into up_out."""
        m = cat.module
        n, h, w, dev = x.n, x.h // 2, x.w // 2, x.buf.device
        c_cur = m[0][1].weight.shape[0]
        pooled = Act(ARENA.get(f"{key}.pool", (n, x.c, h, w), dev), 0, x.c)
        ops.avgpool2(x, pooled)
        self._record(("pool", x, pooled), pooled)
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
        self._record(("up", u, up_out), up_out)

    def run(self, x: Act, out: Act, key: str = "align", retain: Optional[bool] = None) -> Act:
        """x: materialised [N, in_channels, H, W]; out: raw [N, out_channels, H, W].  retain: record the tape run_bwd()
        replays (default: training mode with autograd recording)."""
        s = self.unet
        n, h, w, dev = x.n, x.h, x.w, x.buf.device
        depth = len(self.layer_channels) - 1
        if (h % (1 << depth)) or (w % (1 << depth)):
            raise NotImplementedError(f"alignment U-Net needs H, W divisible by {1 << depth}, got {h}x{w}")
        if retain is None:
            retain = self.training and torch.is_grad_enabled()
        self._tape = [] if retain else None
        self._produced = {}
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
        self._record(("conv", s[5], r2, out), out)
        return out

    # ------------------------------------------------------------------ backward (tape replay)
    def run_bwd(self, out: Act, g_out: Act) -> None:
        """Replay the tape of the last training-mode run() in reverse.  g_out = dL/d(lrelu-read of
        `out`) materialised, i.e. the gradient wrt the activation the NEXT layer read from ``out``
        (its scale/shift/slope describe that read).  Accumulates all parameter gradients."""
        assert self._tape is not None, "run() was not in training mode"
        grads = {}
        counter = [0]

        def tmp(c, h, w, n):
            counter[0] += 1
            return Act(ARENA.get(f"abwd.t{counter[0]}", (n, c, h, w), out.buf.device), 0, c)

        shared = set()          # gradient tensors two readers hold (a residual add hands ONE tensor to both operands)

        def gkey(a: Act):
            return (a.buf.data_ptr(), a.coff, a.c)

        def accum_input(x: Act, g: Act, both: bool = False):
            """g covers x's channels; split it over the activations that produced x's buffer.  ``both``: another reader holds
            the same g (round 6: no private copy per reader -- a sum INTO a shared tensor goes to a fresh one instead, so the
            step has one out-of-place add where it had a copy + an in-place add; same operands, same order, same bits)."""
            ranges = self._produced.get(x.buf.data_ptr())
            if not ranges:
                return                                    # network input: no gradient needed
            for coff, c in ranges:
                if coff >= x.coff and coff + c <= x.coff + x.c:
                    key = (x.buf.data_ptr(), coff, c)
                    piece = g.view(coff - x.coff, c)
                    if key in grads:
                        cur = grads[key]
                        if gkey(cur) in shared:
                            fresh = tmp(c, x.h, x.w, x.n)
                            ops.add(cur, piece, fresh)
                            grads[key] = fresh
                        else:
                            ops.add(cur, piece, cur)
                    else:
                        grads[key] = piece
                        if both:
                            shared.add(gkey(piece))

        def take(a: Act) -> Optional[Act]:
            return grads.get((a.buf.data_ptr(), a.coff, a.c))

        grads[(out.buf.data_ptr(), out.coff, out.c)] = g_out
        for op in reversed(self._tape):
            kind = op[0]
            if kind in ("cba", "conv"):
                if kind == "cba":
                    _, conv, bn, x, o = op
                else:
                    _, conv, x, o = op
                    bn = None
                g = take(o)
                if g is None:
                    continue
                n, h, w = o.n, o.h, o.w
                dy = tmp(o.c, h, w, n)
                if bn is not None and bn.training:
                    # per channel over (N, H, W): dbeta = sum u, dgamma = sum u * yn with yn = (yh - beta)/gamma
                    ops.bn_act_bwd(g, o, bn.weight, bn.bias, _grad_of(bn.weight), _grad_of(bn.bias), dy)
                else:
                    ops.act_bwd(g, o, dy, instance_norm=False)   # eval BN / plain activation read
                    if bn is not None:
                        _bn_eval_param_grads(g, o, bn)
                if bn is not None and bn.training:
                    # a bias in front of train-mode BatchNorm has gradient sum(dy) over (N, H, W) = 0 EXACTLY (the batch
                    # mean is subtracted again); the reference's autograd returns rounding noise there (1e-9-sized, checked
                    # against its fixtures).  Leave the zero the flat gradient buffer holds: two launches per layer less.
                    _grad_of(conv.bias)
                else:
                    ops.bias_grad_acc(ops.plane_stats(dy, tag="abwd.b"), _grad_of(conv.bias))
                ops.conv2d_wgrad(x, dy, _grad_of(conv.weight), accumulate=True)
                if self._produced.get(x.buf.data_ptr()):
                    gx = tmp(x.c, h, w, n)
                    ops.conv2d_dgrad(dy, conv.weight, gx)
                    accum_input(x, gx)
            elif kind == "add":
                _, a, b, o = op
                g = take(o)
                if g is None:
                    continue
                if SHARE_RES_GRADS[0]:
                    accum_input(a, g, both=True)
                    accum_input(b, g, both=True)
                else:                                                 # (SAN_UNET_SHARE_GRADS=0: the round-5 form, same bits)
                    accum_input(a, g)
                    g2 = tmp(o.c, o.h, o.w, o.n)
                    ops.apply(g, g2)                                  # second reader gets its own copy
                    accum_input(b, g2)
            elif kind == "pool":
                _, x, o = op
                g = take(o)
                if g is None:
                    continue
                sc, sh = _const_affine("abwd.q", o.n, o.c, 0.25, o.buf.device)
                gx = tmp(x.c, x.h, x.w, x.n)
                ops.upsample2(Act(g.buf, g.coff, g.c, _expand_aff(sc, g), _expand_aff(sh, g), 1.0), gx)
                accum_input(x, gx)
            elif kind == "up":
                _, u, o = op
                g = take(o)
                if g is None:
                    continue
                sc, sh = _const_affine("abwd.f", u.n, u.c, 4.0, u.buf.device)
                gu = tmp(u.c, u.h, u.w, u.n)
                ops.avgpool2(Act(g.buf, g.coff, g.c, _expand_aff(sc, g), _expand_aff(sh, g), 1.0), gu)
                accum_input(u, gu)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, _, h, w = x.shape
        y = torch.empty((n, self.out_channels, h, w), device=x.device)
        self.run(ops.full(x.contiguous()), ops.full(y))
        return y


def _bn_eval_param_grads(g: Act, o: Act, bn) -> None:
    """Eval-mode BatchNorm (running statistics: a fixed per-channel affine yh = gamma xn + beta): d beta = sum u and
    d gamma = sum u xn = (sum u yh - beta sum u) / gamma over (N, H, W), from the two plane sums of san_plane_dot_stats.
    Rare path (a backward through a network in eval mode): the per-channel arithmetic is a handful of tiny torch ops."""
    part = ops.plane_dot_part(g, o, "bn.eval").double()
    su, suy = part[..., 0].sum(dim=(0, 2)), part[..., 1].sum(dim=(0, 2))
    gamma, beta = bn.weight.detach().double(), bn.bias.detach().double()
    safe = torch.where(gamma != 0, gamma, torch.ones_like(gamma))
    _grad_of(bn.bias).add_(su.float())
    _grad_of(bn.weight).add_(torch.where(gamma != 0, (suy - beta * su) / safe, torch.zeros_like(su)).float())


def _grad_of(p: torch.Tensor) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def _const_affine(name: str, n: int, c: int, value: float, dev):
    sc = ARENA.get(f"{name}.sc{value}", (n, c), dev)
    sh = ARENA.get(f"{name}.sh0", (n, c), dev)
    if not getattr(sc, "_san_filled", False):
        sc.fill_(value)
        sh.zero_()
        sc._san_filled = True
    return sc, sh


def _expand_aff(a: torch.Tensor, g: Act) -> torch.Tensor:
    """Affine arrays are laid out like their buffer's channel axis ([n, ctot], read at coff + ch):
    place a per-view [n, c] constant array at the view's offset of a [n, ctot] array."""
    if g.coff == 0 and g.ctot == a.shape[1]:
        return a
    # (constant arrays: built once per (value array, view) and kept in the arena)
    full = ARENA.get(f"abwd.aff.{a.data_ptr()}.{g.ctot}.{g.coff}.{g.c}", (a.shape[0], g.ctot), a.device)
    if not getattr(full, "_san_filled", False):
        full.zero_()
        full[:, g.coff:g.coff + g.c] = a
        full._san_filled = True
    return full


def _arena_act(name, n, c, h, w, dev) -> Act:
    return Act(ARENA.get(name, (n, c, h, w), dev), 0, c, ARENA.get(name + ".sc", (n, c), dev),
               ARENA.get(name + ".sh", (n, c), dev), SLOPE)


def _running_factors(bn: torch.nn.BatchNorm2d, count: int, count_scale: int):
    """(momentum m, variance factor) of running = (1-m)*running + m*batch (m = 0.1) with the unbiased batch variance.
    ``count_scale`` = 4 for the Up blocks, whose statistics are taken at low resolution: the reference sees every value 4
    times, which changes only the n/(n-1) factor of the unbiased variance."""
    m = bn.momentum if bn.momentum is not None else 0.1
    factor = 1.0
    if count_scale != 1 and count > 1:
        big = count * count_scale
        factor = ((count - 1) / count) * (big / (big - 1))
    return m, factor
