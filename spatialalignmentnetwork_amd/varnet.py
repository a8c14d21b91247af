"""Host-side mirror of the reference's varnet.py, executing on libsan_hip.so.

Same class names, constructor signatures, ``forward`` arguments and state_dict
keys as the reference (varnet.py:24-530), so checkpoints load unchanged and
``model.py``-style callers drop in.  The arithmetic is NOT PyTorch's: every
module below only holds parameters (in torch.nn containers, for identical key
names and default initialisation) and drives the HIP kernels through
``ops`` with the lazy-normalisation scheme described in DESIGN.md:

    conv (raw output + per-tile statistics)  ->  norm_finalize (scale, shift)
    ->  the NEXT kernel applies lrelu(scale*x + shift) while loading.

So InstanceNorm + LeakyReLU never touch memory on their own, concatenations are
zero-copy (producers write at channel offsets) and the NormUnet normalise /
un-normalise steps are folded into the first conv's loader and the last conv's
epilogue.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops
from .ops import Act, GLOBAL_ARENA as ARENA
from .signal_utils import rss as _rss  # noqa: F401  (API parity: reference imports these names here)

IN_EPS = 1e-5
# NormUnet's backward as head + U-Net + tail launches inside a cascade (round 6; SAN_FUSED_NU_BWD=0: the separate launches)
import os as _os
FUSED_NU_BWD = [_os.environ.get("SAN_FUSED_NU_BWD", "1") != "0"]
# InstanceNorm finalisation + average pooling of an encoder level in one launch (round 6; SAN_FUSED_FIN_POOL=0: two launches)
FUSED_FIN_POOL = [_os.environ.get("SAN_FUSED_FIN_POOL", "1") != "0"]
# The cascade-boundary launch emits the next NormUnet's input statistics (round 6; SAN_DC_STATS=0: a san_plane_stats launch per cascade)
DC_STATS = [_os.environ.get("SAN_DC_STATS", "1") != "0"]


class ConvBlock(nn.Module):
    """conv3x3(no bias) -> InstanceNorm -> LeakyReLU(0.2), twice.  Reference: varnet.py:122-156."""

    def __init__(self, in_chans: int, out_chans: int):
        super().__init__()
        self.in_chans, self.out_chans = in_chans, out_chans
        self.layers = nn.Sequential(
            nn.Conv2d(in_chans, out_chans, kernel_size=3, padding=1, bias=False),
            nn.InstanceNorm2d(out_chans),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
            nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False),
            nn.InstanceNorm2d(out_chans),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
        )

    def run(self, x: Act, mid: Act, out: Act, tag: str, pooled: Optional[Act] = None) -> Act:
        """x -> mid (raw) -> out (raw); fills the scale/shift of mid and out.  pooled: also avg_pool2d of the block's activated
        output (the encoder levels, varnet.py:95-99) -- in the SAME launch as the second InstanceNorm's finalisation where that
        is a launch (ops.norm_finalize_pool), else by avgpool2."""
        # (split-K layers -- deep K on small planes -- finalise the InstanceNorm affine in their reduction pass: None)
        part = ops.conv2d(x, self.layers[0].weight, None, mid, stats=True, tag=tag, instance_norm_eps=IN_EPS)
        if part is not None:
            ops.norm_finalize(part, ops.NORM_INSTANCE, IN_EPS, mid.scale, mid.shift, mid.coff)
        part = ops.conv2d(mid, self.layers[3].weight, None, out, stats=True, tag=tag, instance_norm_eps=IN_EPS)
        if part is not None and pooled is not None and FUSED_FIN_POOL[0]:
            ops.norm_finalize_pool(part, IN_EPS, out, pooled)
            return out
        if part is not None:
            ops.norm_finalize(part, ops.NORM_INSTANCE, IN_EPS, out.scale, out.shift, out.coff)
        if pooled is not None:
            ops.avgpool2(out, pooled)
        return out

    def run_bwd(self, g_out: Act, x: Act, mid: Act, out: Act, g_in: Optional[Act], g_pooled: Optional[Act] = None) -> None:
        """g_out = dL/d(lrelu(IN(out))) (materialised).  Accumulates both weight gradients and,
        if g_in is given, writes dL/d(T(x)) (gradient wrt the lazily activated block input).
        g_pooled: the gradient wrt avg_pool2d(block output), added on the fly as 0.25 * g_pooled(p // 2)."""
        n, h, w, dev = x.n, x.h, x.w, x.buf.device
        wa, wb = self.layers[0].weight, self.layers[3].weight
        dyb = Act(ops.wgrad_dy_buffer("bwd.dy", (n, out.c, h, w), dev, ARENA), 0, out.c)
        ops.act_bwd(g_out, out, dyb, instance_norm=True, g2=g_pooled)
        ops.conv2d_wgrad(mid, dyb, _grad_of(wb), accumulate=True)
        g_mid = Act(ARENA.get("bwd.gmid", (n, mid.c, h, w), dev), 0, mid.c)
        ops.conv2d_dgrad(dyb, wb, g_mid)
        dya = Act(ops.wgrad_dy_buffer("bwd.dy", (n, mid.c, h, w), dev, ARENA), 0, mid.c)
        ops.act_bwd(g_mid, mid, dya, instance_norm=True)
        ops.conv2d_wgrad(x, dya, _grad_of(wa), accumulate=True)
        if g_in is not None:
            ops.conv2d_dgrad(dya, wa, g_in)

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        n, _, h, w = image.shape
        dev = image.device
        mid = _new_act(n, self.out_chans, h, w, dev, 0.2)
        out = _new_act(n, self.out_chans, h, w, dev, 0.2)
        self.run(ops.full(image.contiguous()), mid, out, "cb")
        y = torch.empty_like(out.buf)
        ops.apply(out, ops.full(y))
        return y


class TransposeConvBlock(nn.Module):
    """ConvTranspose2d 2x2 s2 (no bias) -> InstanceNorm -> LeakyReLU(0.2).  Reference: varnet.py:159-192."""

    def __init__(self, in_chans: int, out_chans: int):
        super().__init__()
        self.in_chans, self.out_chans = in_chans, out_chans
        self.layers = nn.Sequential(
            nn.ConvTranspose2d(in_chans, out_chans, kernel_size=2, stride=2, bias=False),
            nn.InstanceNorm2d(out_chans),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
        )

    def run(self, x: Act, out: Act, tag: str, also: Optional[Act] = None) -> Act:
        """``also``: a second view whose (scale, shift) receive the same InstanceNorm affine (the reflect-padded copy
        of ``out`` inside the concatenation buffer, Unet.run)."""
        part = ops.tconv2x2(x, self.layers[0].weight, out, stats=True, tag=tag)
        ops.norm_finalize(part, ops.NORM_INSTANCE, IN_EPS, out.scale, out.shift, out.coff)
        if also is not None:
            ops.norm_finalize(part, ops.NORM_INSTANCE, IN_EPS, also.scale, also.shift, also.coff)
        return out

    def run_bwd(self, g_out: Act, x: Act, out: Act, g_in: Act) -> None:
        """Backward of tconv + IN + LReLU.  The transposed conv is a 1x1 conv to 4*Cout virtual
        channels (tap-major) + pixel shuffle, so after un-shuffling dy both gradients are plain
        1x1-conv gradients: dgrad with the weight viewed as [Cin, 4*Cout, 1, 1]."""
        wt = self.layers[0].weight                       # [Cin, Cout, 2, 2]
        cin, cout = wt.shape[0], wt.shape[1]
        n, h, w, dev = x.n, x.h, x.w, x.buf.device
        dyp = Act(ops.wgrad_dy_buffer("bwd.dyp", (n, 4 * cout, h, w), dev, ARENA), 0, 4 * cout)
        if ops.act_bwd_unshuffle_ok(out) and g_out.buf.data_ptr() % 16 == 0:
            # the activation backward writes dy pixel-unshuffled itself (no second pass over the tensor)
            ops.act_bwd_ex(g_out, out, dyp, instance_norm=True, unshuffle=True)
        else:
            dyt = Act(ARENA.get("bwd.dyt", (n, cout, 2 * h, 2 * w), dev), 0, cout)
            ops.act_bwd(g_out, out, dyt, instance_norm=True)
            ops.unshuffle2(dyt, dyp)
        wv = _view_cached(wt, (cin, 4 * cout, 1, 1))
        # dL/dx[ci] = sum_c' Wv[ci, c'] dy'[c']  == forward 1x1 conv with weight [cout'=Cin, cin'=4Cout]
        ops.conv2d(dyp, wv, None, g_in, grad_input=True)
        if ops.wgrad1x1_bf16x3_ok(x, dyp):
            # the reduction writes [Cin][4 Cout] = the ConvTranspose2d weight layout: accumulate in place
            ops.conv2d_wgrad1x1_bf16x3(x, dyp, _grad_of(wt), accumulate=True, transposed=True)
        else:
            dwv = ARENA.get("bwd.dwv", (4 * cout, cin, 1, 1), dev)
            ops._conv2d_wgrad(x, dyp, dwv, accumulate=False, scratch_tag=".inline")      # in line (own scratch): its result is consumed right here
            gw = _grad_of(wt)
            ops._lib.rec(lambda: gw.add_(dwv.view(4 * cout, cin).t().reshape(cin, cout, 2, 2)))

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        n, _, h, w = image.shape
        out = _new_act(n, self.out_chans, 2 * h, 2 * w, image.device, 0.2)
        self.run(ops.full(image.contiguous()), out, "tb")
        y = torch.empty_like(out.buf)
        ops.apply(out, ops.full(y))
        return y


def _grad_of(p: torch.Tensor) -> torch.Tensor:
    """The parameter's .grad buffer (zero-initialised on first use, accumulated into afterwards)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def _view_cached(w: torch.Tensor, shape) -> torch.Tensor:
    """A reshaped view of a parameter, made ONCE per storage: the packing registry keys its jobs by tensor object, and a view
    re-made whenever the parameter's version moved (the restore after a recording's warm-up copies every parameter) was a NEW job
    -- packed by a launch of its own, 48 of them in every recorded step (round 6: 585 single packs in a 13-step profile)."""
    key = (w.data_ptr(), tuple(shape))
    hit = getattr(w, "_san_view", None)
    if hit is None or hit[0] != key:
        hit = (key, w.detach().view(*shape))
        w._san_view = hit
    return hit[1]


def _const_affine(name: str, n: int, c: int, value: float, dev):
    """[n, c] scale filled with `value` and a zero shift (lazy 'multiply by constant')."""
    sc = ARENA.get(f"{name}.sc{value}", (n, c), dev)
    sh = ARENA.get(f"{name}.sh0", (n, c), dev)
    if not getattr(sc, "_san_filled", False):
        sc.fill_(value)
        sh.zero_()
        sc._san_filled = True
    return sc, sh


def _new_act(n, c, h, w, dev, slope) -> Act:
    return Act(torch.empty((n, c, h, w), device=dev), 0, c, torch.empty((n, c), device=dev),
               torch.empty((n, c), device=dev), slope)


def _arena_act(name, n, c, h, w, dev, slope) -> Act:
    return Act(ARENA.get(name, (n, c, h, w), dev), 0, c, ARENA.get(name + ".sc", (n, c), dev),
               ARENA.get(name + ".sh", (n, c), dev), slope)


class Unet(nn.Module):
    """fastMRI U-Net.  Reference: varnet.py:24-119."""

    def __init__(self, in_chans: int, out_chans: int, chans: int = 32, num_pool_layers: int = 4):
        super().__init__()
        self.in_chans, self.out_chans, self.chans, self.num_pool_layers = in_chans, out_chans, chans, num_pool_layers
        self.down_sample_layers = nn.ModuleList([ConvBlock(in_chans, chans)])
        ch = chans
        for _ in range(num_pool_layers - 1):
            self.down_sample_layers.append(ConvBlock(ch, ch * 2))
            ch *= 2
        self.conv = ConvBlock(ch, ch * 2)
        self.up_conv = nn.ModuleList()
        self.up_transpose_conv = nn.ModuleList()
        for _ in range(num_pool_layers - 1):
            self.up_transpose_conv.append(TransposeConvBlock(ch * 2, ch))
            self.up_conv.append(ConvBlock(ch * 2, ch))
            ch //= 2
        self.up_transpose_conv.append(TransposeConvBlock(ch * 2, ch))
        self.up_conv.append(nn.Sequential(ConvBlock(ch * 2, ch), nn.Conv2d(ch, self.out_chans, kernel_size=1, stride=1)))
        self._tapes = {}          # key -> activations retained by run() for run_bwd()

    def run(self, x: Act, out: Act, out_scale: Optional[torch.Tensor] = None,
            out_shift: Optional[torch.Tensor] = None, key: str = "unet") -> Act:
        """x: lazily normalised input view; out: destination view (materialised,
        optionally through the per-(n, c) output affine).  All intermediates
        come from the shared arena (inference: cascades reuse them)."""
        P = self.num_pool_layers
        n, h, w, dev = x.n, x.h, x.w, x.buf.device
        tape = {"x": x, "blocks": [], "pooled": [], "ups": []}
        if (h >> P) < 1 or (w >> P) < 1:
            raise ValueError(f"U-Net input {h}x{w} is too small for {P} pooling levels")
        tagp = f"{key}.c{self.chans}"
        cur = x
        cats = []
        ch = self.chans
        hh, ww = h, w
        for i in range(P):
            cat = _arena_act(f"{tagp}.cat{i}", n, 2 * ch, hh, ww, dev, 0.2)   # [0,ch): up path, [ch,2ch): skip
            mid = _arena_act(f"{tagp}.mid{i}", n, ch, hh, ww, dev, 0.2)
            pooled = Act(ARENA.get(f"{tagp}.pool{i}", (n, ch, hh // 2, ww // 2), dev), 0, ch)
            odd = bool((hh | ww) & 1)
            self.down_sample_layers[i].run(cur, mid, cat.view(ch, ch), tagp, pooled=None if odd else pooled)
            tape["blocks"].append((cur, mid, cat.view(ch, ch)))
            cats.append(cat)
            if odd:
                # F.avg_pool2d drops the odd last row / column (varnet.py:99): activate + crop, then pool
                even = Act(ARENA.get(f"{tagp}.even{i}", (n, ch, hh & ~1, ww & ~1), dev), 0, ch)
                ops.window_copy(cat.view(ch, ch), even)
                ops.avgpool2(even, pooled)
            tape["pooled"].append(pooled)
            cur = pooled
            hh, ww, ch = hh // 2, ww // 2, ch * 2
        mid = _arena_act(f"{tagp}.midb", n, ch, hh, ww, dev, 0.2)
        bot = _arena_act(f"{tagp}.bot", n, ch, hh, ww, dev, 0.2)
        tape["bott"] = (cur, mid, bot)
        cur = self.conv.run(cur, mid, bot, tagp)
        for i in range(P):
            ch //= 2
            lvl = P - 1 - i
            cat = cats[lvl]
            hh, ww = cat.h, cat.w
            tmp = None
            if (2 * cur.h, 2 * cur.w) != (hh, ww):
                # odd size at this level: the transposed conv's output is one row / column short and the reference
                # reflect-pads it on the bottom / right AFTER its InstanceNorm + LeakyReLU (varnet.py:107-114).  Both
                # are per-channel maps, so the RAW values are reflected and the affine is shared.
                tmp = _arena_act(f"{tagp}.tpad{lvl}", n, ch, 2 * cur.h, 2 * cur.w, dev, 0.2)
                self.up_transpose_conv[i].run(cur, tmp, tagp, also=cat.view(0, ch))
                ops.window_copy(Act(tmp.buf, 0, ch), Act(cat.buf, 0, ch), mode=1)
            else:
                self.up_transpose_conv[i].run(cur, cat.view(0, ch), tagp)
            mid = _arena_act(f"{tagp}.umid{lvl}", n, ch, hh, ww, dev, 0.2)
            up = _arena_act(f"{tagp}.up{lvl}", n, ch, hh, ww, dev, 0.2)
            block = self.up_conv[i] if i < P - 1 else self.up_conv[i][0]
            tape["ups"].append((cur, cat, mid, up, tmp))      # cur = the transposed conv's input
            cur = block.run(cat, mid, up, tagp)
        last = self.up_conv[P - 1][1]
        ops.conv2d(cur, last.weight, last.bias, out, stats=False, out_scale=out_scale, out_shift=out_shift)
        tape["last_in"] = cur
        self._tapes[key] = tape
        return out

    def run_bwd(self, g_conv: torch.Tensor, key: str = "unet", bias_grad: bool = True,
                input_grad: bool = True) -> Optional[torch.Tensor]:
        """Backward of the last ``run(..., key=key)``.  g_conv [N, out_chans, H, W] is the gradient
        wrt the final 1x1 conv's output BEFORE the optional output affine.  Accumulates every
        parameter gradient and returns dL/d(T(x)) [N, in_chans, H, W] for the lazily read input.
        bias_grad=False: the caller adds the last convolution's bias gradient itself (NormUnet's fused tail);
        input_grad=False: nobody reads dL/d(input) (the sensitivity network's input is data): the first convolution's data
        gradient is not computed and None is returned."""
        tape = self._tapes[key]
        P = self.num_pool_layers
        x = tape["x"]
        n, dev = x.n, x.buf.device
        last = self.up_conv[P - 1][1]
        cur = tape["last_in"]
        gc = ops.full(g_conv.contiguous())
        # final 1x1 conv (+bias): bias gradient = plane sums of g
        if bias_grad:
            ops.bias_grad_acc(ops.plane_stats(gc, tag="bgrad"), _grad_of(last.bias))
        ops.conv2d_wgrad(cur, gc, _grad_of(last.weight), accumulate=True)
        g = Act(ARENA.get(f"bwd.g.{cur.c}.{cur.h}.{cur.w}", (n, cur.c, cur.h, cur.w), dev), 0, cur.c)
        ops.conv2d_dgrad(gc, last.weight, g)
        skip_g = [None] * P
        for i in reversed(range(P)):
            tin, cat, mid, up, tmp = tape["ups"][i]
            lvl = P - 1 - i
            ch = up.c
            block = self.up_conv[i] if i < P - 1 else self.up_conv[i][0]
            g_cat = Act(ARENA.get(f"bwd.gcat{lvl}", (n, 2 * ch, cat.h, cat.w), dev), 0, 2 * ch)
            block.run_bwd(g, cat, mid, up, g_cat)
            skip_g[lvl] = g_cat.view(ch, ch)
            g = Act(ARENA.get(f"bwd.g.{tin.c}.{tin.h}.{tin.w}", (n, tin.c, tin.h, tin.w), dev), 0, tin.c)
            if tmp is None:
                self.up_transpose_conv[i].run_bwd(g_cat.view(0, ch), tin, cat.view(0, ch), g)
            else:                                            # adjoint of the reflect pad, then the unpadded block
                g_tmp = Act(ARENA.get(f"bwd.gtpad{lvl}", (n, ch, tmp.h, tmp.w), dev), 0, ch)
                ops.window_copy(g_cat.view(0, ch), g_tmp, mode=2)
                self.up_transpose_conv[i].run_bwd(g_tmp, tin, tmp, g)
        bx, bmid, bot = tape["bott"]
        g_pool = Act(ARENA.get(f"bwd.gp.{bx.c}.{bx.h}.{bx.w}", (n, bx.c, bx.h, bx.w), dev), 0, bx.c)
        self.conv.run_bwd(g, bx, bmid, bot, g_pool)
        for i in reversed(range(P)):
            bin_, bmid_, bout = tape["blocks"][i]
            ch = bout.c
            # avg-pool backward (x0.25, nearest up-sampling) + the skip connection's gradient
            want_in = input_grad or i > 0
            if ops.act_bwd_up_ok(bout):                      # summed inside the block's first activation-backward kernel
                g_next = Act(ARENA.get(f"bwd.gp.{bin_.c}.{bin_.h}.{bin_.w}", (n, bin_.c, bin_.h, bin_.w), dev), 0, bin_.c) if want_in else None
                self.down_sample_layers[i].run_bwd(skip_g[i], bin_, bmid_, bout, g_next, g_pooled=Act(g_pool.buf, 0, ch))
                g_pool = g_next
                continue
            sc, sh = _const_affine("bwd.quarter", n, ch, 0.25, dev)
            up = Act(ARENA.get(f"bwd.pup{i}", (n, ch, bout.h, bout.w), dev), 0, ch)
            if (bout.h | bout.w) & 1:                         # the pooled-away odd row / column gets no gradient
                upe = Act(ARENA.get(f"bwd.pupe{i}", (n, ch, bout.h & ~1, bout.w & ~1), dev), 0, ch)
                ops.upsample2(Act(g_pool.buf, 0, ch, sc, sh, 1.0), upe)
                ops.window_copy(upe, up)
            else:
                ops.upsample2(Act(g_pool.buf, 0, ch, sc, sh, 1.0), up)
            ops.add(up, skip_g[i], up)
            g_pool = Act(ARENA.get(f"bwd.gp.{bin_.c}.{bin_.h}.{bin_.w}", (n, bin_.c, bin_.h, bin_.w), dev), 0, bin_.c) if want_in else None
            self.down_sample_layers[i].run_bwd(up, bin_, bmid_, bout, g_pool)
        return g_pool.buf if g_pool is not None else None

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        assert not torch.is_complex(image)
        n, _, h, w = image.shape
        y = torch.empty((n, self.out_chans, h, w), device=image.device)
        self.run(ops.full(image.contiguous()), ops.full(y))
        return y


class NormUnet(nn.Module):
    """Normalised U-Net on complex [B, 1, H, W] data.  Reference: varnet.py:200-332."""

    def __init__(self, chans: int, num_pools: int, in_chans: int = 1, out_chans: int = 1, use_ref: bool = False):
        super().__init__()
        if in_chans != 1 or out_chans != 1:
            # the reference's norm() broadcasts (B,2c,H,W) against (B,2,1,1): only c == 1 works there too
            raise NotImplementedError("NormUnet is only defined for in_chans == out_chans == 1")
        self.use_ref = use_ref
        self.unet = Unet(in_chans=in_chans * (3 if use_ref else 2), out_chans=out_chans * 2, chans=chans,
                         num_pool_layers=num_pools)
        if use_ref:
            self.ref_norm = nn.InstanceNorm2d(in_chans)
        self.in_chans, self.out_chans = in_chans, out_chans
        self._tapes = {}

    # -- fused path -------------------------------------------------------
    def input_buffer(self, b: int, h: int, w: int, dev, key: str) -> Act:
        """[B, 2 or 3, H, W] buffer the k-space kernels write the planar image into."""
        c = 3 if self.use_ref else 2
        return _arena_act(f"{key}.nu_in{c}", b, c, h, w, dev, 1.0)

    def set_ref(self, xin: Act, ref: torch.Tensor) -> None:
        """Place ref (real [B,1,H,W]) as channel 2 with its InstanceNorm affine
        (varnet.py:315-319).  The same ref feeds every cascade, so this runs once."""
        ops.apply(ops.full(ref), xin.view(2, 1))
        part = ops.plane_stats(xin.view(2, 1), tag="ref")
        ops.norm_finalize(part, ops.NORM_INSTANCE, IN_EPS, xin.scale, xin.shift, 2)

    @staticmethod
    def pad_sizes(h: int, w: int):
        """(top, left, padded h, padded w): zero padding to multiples of 16, the smaller half first (varnet.py:275-289)."""
        hp, wp = ((h - 1) | 15) + 1, ((w - 1) | 15) + 1
        return (hp - h) // 2, (wp - w) // 2, hp, wp

    def run(self, xin: Act, out_planar: torch.Tensor, key: str, part: Optional[torch.Tensor] = None) -> torch.Tensor:
        """xin channels 0,1 hold the planar complex image (raw).  Writes the
        un-normalised planar output [B,2,H,W].  ``part``: (count, mean, M2) records [B, 2, tiles, 3] of the two planes that the
        producer of xin already emitted (ops.dc_rows(m_stats=)); None: one san_plane_stats launch here."""
        b, h, w = xin.n, xin.h, xin.w
        dev = xin.buf.device
        # [0] = std / mean of the planes (the un-normalisation affine), [1] = guarded 1 / std and -mean / std (the backward's
        # affine that recovers the U-Net output from the un-normalised one): one launch writes all four
        std2 = ARENA.get(f"{key}.gn_std", (2, b, 2), dev)
        mean2 = ARENA.get(f"{key}.gn_mean", (2, b, 2), dev)
        std, mean = std2[0], mean2[0]
        if part is None:
            part = ops.plane_stats(xin.view(0, 2), tag="gn")
        ops.norm_finalize(part, ops.NORM_GROUP_BWD, 1e-6, xin.scale, xin.shift, 0, aux_a=std2, aux_b=mean2)
        top, left, hp, wp = self.pad_sizes(h, w)
        if (hp, wp) == (h, w):
            self.unet.run(xin, ops.full(out_planar), out_scale=std, out_shift=mean, key=key)
        else:
            # NormUnet.pad (varnet.py:275-289, 311-319): the NORMALISED image and the normalised reference are zero-padded,
            # the U-Net runs on the padded planes, the result is cropped (unpad, :291-299) and then un-normalised
            xpad = Act(ARENA.get(f"{key}.nu_pad", (b, xin.c, hp, wp), dev), 0, xin.c)
            ops.window_copy(xin, xpad, top, left)
            upad = ARENA.get(f"{key}.nu_upad", (b, 2, hp, wp), dev)
            self.unet.run(xpad, ops.full(upad), key=key)
            ops.window_copy(Act(upad, 0, 2, std, mean, 1.0), ops.full(out_planar), -top, -left)
        self._tapes[key] = (xin, out_planar, std, mean, std2[1], mean2[1])
        return out_planar

    def run_bwd(self, g_out: torch.Tensor, key: str, want_ref_grad: bool = False, g_ref_acc: Optional[torch.Tensor] = None,
                input_grad: bool = True, fuse: Optional[dict] = None):
        """Backward of the last run(key).  g_out: dL/d(output) planar [B,2,H,W].  g_ref_acc: a tensor dL/d(ref) is ADDED to
        (then also the returned one).  Returns
        (dL/d(input planar) [B,2,H,W], dL/d(ref) [B,1,H,W] or None); parameter gradients accumulate.
        input_grad=False: the input is data (the sensitivity network): (None, None) is returned and nothing input-side is computed.
        fuse (VarNetBlock.run_bwd_img): dict(gd, sens, gS, r, t1, xs, dcw=(partials, grad)) -- what the caller would do with
        dL/d(input) next (gd += g_m S, the sensitivity-map accumulation) and dc_weight's pending gradient partials: done by the
        LAST launch of this backward (ops.normunet_bwd_tail) without storing g_m; the first value returned is then None.

        With x^ = (m - mu)/(sigma + eps) (sigma = unbiased std of the plane), U = unet(x^, ref^) and
        out = U*sigma + mu:
            dL/dU   = g_out * sigma
            dmu     = sum(g_out)   - sum(g_x^)/(sigma+eps)
            dsigma  = sum(g_out*U) - sum(g_x^ * x^)/(sigma+eps)
            dL/dm   = g_x^/(sigma+eps) + dmu/n + dsigma * (m - mu)/((n-1)*sigma)."""
        xin, out_planar, std, mean, isd, nshift = self._tapes[key]
        b, h, w, dev = xin.n, xin.h, xin.w, xin.buf.device
        nel = h * w
        g_out = g_out.contiguous()
        top, left, hp, wp = self.pad_sizes(h, w)
        padded = (hp, wp) != (h, w)
        fused = fuse is not None and not padded and FUSED_NU_BWD[0]
        zero_sh = ARENA.get("bwd.zero_sh", tuple(std.shape), dev, zero=True)
        # a constant plane (e.g. an all-zero slice) has std == 0: U cannot be recovered from U*0 + mean, and torch's
        # std backward gives that plane no d sigma term at all (isd = 0 there, written by the forward's finalisation;
        # san_normunet_bwd_coefs does the same)
        part_b = None
        if fused:
            g_u = ops.wgrad_dy_buffer("bwd.g_u", (b, 2, h, w), dev, ARENA)
            part_b = ops.normunet_bwd_head(g_out, out_planar, isd, nshift, std, g_u)      # one pass: g_u and the chunk sums
            g_xh = self.unet.run_bwd(g_u, key, bias_grad=False)                          # [B, 2 or 3, H, W]
        else:
            if input_grad:
                part_b = ops.plane_dot_part(ops.full(g_out), Act(out_planar, 0, 2, isd, nshift, 1.0), "nu.b")
            if not padded:
                g_u = ops.wgrad_dy_buffer("bwd.g_u", (b, 2, h, w), dev, ARENA)
                ops.apply(Act(g_out, 0, 2, std, zero_sh, 1.0), ops.full(g_u))
                g_xh = self.unet.run_bwd(g_u, key, input_grad=input_grad)
            else:
                g_u = ops.wgrad_dy_buffer("bwd.g_u", (b, 2, hp, wp), dev, ARENA)
                ops.window_copy(Act(g_out, 0, 2, std, zero_sh, 1.0), ops.full(g_u), top, left)     # adjoint of the crop
                g_pad = self.unet.run_bwd(g_u, key, input_grad=input_grad)
                if input_grad:
                    g_xh = ARENA.get("bwd.g_xh_crop", (b, xin.c, h, w), dev)
                    ops.window_copy(ops.full(g_pad), ops.full(g_xh), -top, -left)                  # adjoint of the zero pad
        if not input_grad:
            assert fuse is None and not want_ref_grad
            return None, None
        ref = self.use_ref and want_ref_grad
        if fused:
            g_ref = None
            if ref:
                g_ref = g_ref_acc if g_ref_acc is not None else torch.empty((b, 1, h, w), device=dev)
            xc = 3 if ref else 2
            part_x = ops.plane_dot_part(Act(g_xh, 0, xc), xin.view(0, xc), "nu.a")
            last = self.unet.up_conv[self.unet.num_pool_layers - 1][1]
            dcw_part, dcw_grad = fuse.get("dcw") or (None, None)
            ops.normunet_bwd_tail(part_b, part_x, xin, std, nel, g_xh, fuse["gd"], fuse["sens"], fuse.get("gS"), fuse.get("r"),
                                  fuse.get("t1"), fuse.get("xs"), -1.0, g_ref, ref and g_ref_acc is not None, _grad_of(last.bias),
                                  dcw_part, -1.0, dcw_grad)
            return None, g_ref
        part_a = ops.plane_dot_part(Act(g_xh, 0, 2), xin.view(0, 2), "nu.a")
        a_sc, a_sh, m_sc, m_sh = ops.normunet_bwd_coefs(part_b, part_a, xin, std, nel, g_xh.shape[1])
        g_m = torch.empty((b, 2, h, w), device=dev)
        ops.add(Act(g_xh, 0, 2, a_sc, a_sh, 1.0), Act(xin.buf, xin.coff, 2, m_sc, m_sh, 1.0), ops.full(g_m))
        g_ref = None
        if ref:
            if g_ref_acc is not None:
                # dL/d ref is the sum over the cascades (every cascade reads the same ref): accumulated in place by the kernel
                g_ref = g_ref_acc
                ops.act_bwd_ex(Act(g_xh, 2, 1), xin.view(2, 1), ops.full(g_ref), instance_norm=True, accumulate=True)
            else:
                g_ref = torch.empty((b, 1, h, w), device=dev)
                ops.act_bwd(Act(g_xh, 2, 1), xin.view(2, 1), ops.full(g_ref), instance_norm=True)
        if fuse is not None:                    # (the padded form: what the fused tail would have done, as separate launches)
            ops.sens_grad_prop(fuse.get("gS"), fuse.get("r"), fuse.get("t1"), fuse.get("xs"), g_m, fuse["gd"], fuse["sens"])
            if fuse.get("dcw"):
                ops.partials_add(fuse["dcw"][0], -1.0, fuse["dcw"][1])
            return None, g_ref
        return g_m, g_ref

    # -- reference-compatible entry --------------------------------------
    def forward(self, x: torch.Tensor, ref: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert x.dim() == 4 and torch.is_complex(x) and x.shape[1] == self.in_chans
        b, _, h, w = x.shape
        xin = self.input_buffer(b, h, w, x.device, "nu")
        xr = torch.view_as_real(x.contiguous())
        # complex -> planar through the element-wise materialiser (two strided views are not
        # contiguous, so go through rss-free explicit copies of the re / im planes)
        planar = xr.permute(0, 1, 4, 2, 3).reshape(b, 2, h, w).contiguous()
        ops.apply(ops.full(planar), xin.view(0, 2))
        if self.use_ref:
            assert ref is not None and not torch.is_complex(ref)
            self.set_ref(xin, ref.contiguous())
        else:
            assert ref is None
        out = torch.empty((b, 2, h, w), device=x.device)
        self.run(xin, out, "nu")
        return torch.complex(out[:, 0:1], out[:, 1:2])


class SensitivityModel(nn.Module):
    """Coil sensitivity estimation.  Reference: varnet.py:335-420."""

    def __init__(self, chans: int, num_pools: int, in_chans: int = 1, out_chans: int = 1, mask_center: bool = True):
        super().__init__()
        self.mask_center = mask_center
        self.norm_unet = NormUnet(chans, num_pools, in_chans=in_chans, out_chans=out_chans)

    @staticmethod
    def acs_window(width: int, num_low_frequencies: int, device) -> torch.Tensor:
        """ones on [0, nlf) rolled by (-nlf)//2.  Host-side, cached per (W, nlf).  varnet.py:395-397."""
        key = (width, num_low_frequencies, str(device))
        hit = _ACS_CACHE.get(key)
        if hit is None:
            m = torch.ones(width)
            m[num_low_frequencies:] = 0
            hit = torch.roll(m, (-num_low_frequencies) // 2).to(device)
            _ACS_CACHE[key] = hit
        return hit

    def forward(self, masked_kspace: torch.Tensor, num_low_frequencies: int) -> torch.Tensor:
        n, c, h, w = masked_kspace.shape
        dev = masked_kspace.device
        acs = self.acs_window(w, num_low_frequencies, dev)
        xin = self.norm_unet.input_buffer(n * c, h, w, dev, "sens")
        ops.ifft2c_planar(masked_kspace, acs, xin.buf)
        est = ARENA.get("sens.est", (n * c, 2, h, w), dev)
        self.norm_unet.run(xin, est, "sens")
        self._tape = (est, n, c)
        return ops.sens_normalize(est, n, c)

    def backward(self, g_sens: torch.Tensor) -> None:
        """dL/d(sens maps) -> parameter gradients (the masked k-space input needs none)."""
        est, n, c = self._tape
        g_est = ops.sens_normalize_bwd(est, g_sens)
        self.norm_unet.run_bwd(g_est, "sens", want_ref_grad=False, input_grad=False)     # (the masked k-space is data)


_ACS_CACHE = {}


class VarNetBlock(nn.Module):
    """One cascade: soft data consistency + regulariser.  Reference: varnet.py:488-530."""

    def __init__(self, model: nn.Module):
        super().__init__()
        self.model = model
        self.dc_weight = nn.Parameter(torch.ones(1))

    def sens_expand(self, image: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        """fft2(image * sens_maps): image complex [N,1,H,W], sens_maps complex [N,C,H,W] (varnet.py:508-509).  Runs on
        the fused expand + DC kernel with k = k0 = 0, which yields -fft2(r * S): fed with r = -image (negation is
        exact) it returns the reference's value."""
        n, c, h, w = sens_maps.shape
        assert image.shape == (n, 1, h, w) and torch.is_complex(image)
        dev = image.device
        planar = torch.view_as_real(image.contiguous()).permute(0, 1, 4, 2, 3).reshape(n, 2, h, w).contiguous()
        neg_sc, neg_sh = _const_affine("bwd.neg", n, 2, -1.0, dev)
        r = torch.empty_like(planar)
        ops.apply(Act(planar, 0, 2, neg_sc, neg_sh, 1.0), ops.full(r))
        zeros = ARENA.get("bwd.zero_k", (n, c, h, w), dev, dtype=torch.complex64, zero=True)
        out = torch.empty((n, c, h, w), device=dev, dtype=torch.complex64)
        ones = ARENA.get("expand.ones", (w,), dev)
        ones.fill_(1.0)
        ops.sens_expand_dc(r, sens_maps.contiguous(), zeros, zeros, ones, self.dc_weight, out)
        return out

    def sens_reduce(self, kspace: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        n, c, h, w = kspace.shape
        out = torch.empty((n, 2, h, w), device=kspace.device)
        ops.sens_reduce(kspace.contiguous(), sens_maps.contiguous(), out)
        return torch.complex(out[:, 0:1], out[:, 1:2])

    def run(self, k: torch.Tensor, k0: torch.Tensor, mask_f: torch.Tensor, sens: torch.Tensor, xin: Act,
            k_out: torch.Tensor, key: str, k_cols: Optional[torch.Tensor] = None,
            next_cols: Optional[torch.Tensor] = None) -> torch.Tensor:
        """k_cols: inverse column transform of k left behind by the previous cascade (skips this
        cascade's first FFT pass); next_cols: buffer that receives the same for k_out."""
        n, c, h, w = k.shape
        ops.sens_reduce(k, sens, xin.buf, cols=k_cols)
        r = ARENA.get(f"{key}.r", (n, 2, h, w), k.device)
        self.model.run(xin, r, key)
        ops.sens_expand_dc(r, sens, k, k0, mask_f, self.dc_weight, k_out, next_cols=next_cols)
        self._tape = (k, k0, mask_f, sens, r, key)
        return k_out

    def run_bwd(self, g_kout: torch.Tensor, g_sens: Optional[torch.Tensor], want_ref_grad: bool):
        """Backward of the last run().  g_kout = dL/dk' (complex).  Returns (dL/dk, dL/d ref or None);
        accumulates parameter gradients and, if g_sens is given, the sensitivity-map gradient.

          R = fft2(r*S):           dL/dr = -sum_c conj(S_c) ifft2(g)_c          = -sens_reduce(g, S)
          m = sum_c ifft2(k)_c conj(S_c), r = NormUnet(m):  dL/dk += fft2(g_m * S)
          soft DC:                 dL/dk += g * (1 - w*M);  dL/dw = -sum M Re(conj(g) (k - k0))
        so dL/dk = sens_expand_dc(r := -g_m, S, k := g, k0 := 0): the forward kernel again."""
        k, k0, mask_f, sens, r, key = self._tape
        n, c, h, w = k.shape
        dev = k.device
        hbuf = ARENA.get("bwd.h", (n, 2, h, w), dev)
        ops.sens_reduce(g_kout, sens, hbuf)
        neg_sc, neg_sh = _const_affine("bwd.neg", n, 2, -1.0, dev)
        g_r = ARENA.get("bwd.g_r", (n, 2, h, w), dev)
        ops.apply(Act(hbuf, 0, 2, neg_sc, neg_sh, 1.0), ops.full(g_r))
        g_m, g_ref = self.model.run_bwd(g_r, key, want_ref_grad)
        if g_sens is not None:
            t1 = ops.fft2c(g_kout, inverse=True)          # ifft2(g); the R-term carries sign -1
            x = ops.fft2c(k, inverse=True)
            ops.sens_grad_acc(g_sens, r, t1, x, g_m, -1.0)
        _grad_of(self.dc_weight).add_(ops.dc_weight_grad(g_kout, k, k0, mask_f))
        neg_gm = ARENA.get("bwd.neg_gm", (n, 2, h, w), dev)
        ops.apply(Act(g_m, 0, 2, neg_sc, neg_sh, 1.0), ops.full(neg_gm))
        zeros = ARENA.get("bwd.zero_k", (n, c, h, w), dev, dtype=torch.complex64, zero=True)
        g_k = torch.empty_like(g_kout)
        ops.sens_expand_dc(neg_gm, sens, g_kout, zeros, mask_f, self.dc_weight, g_k)
        return g_k, g_ref

    # -- image-domain form (what VarNet.forward runs) --------------------------------------------
    def run_img(self, x: torch.Tensor, k0x: torch.Tensor, mask_f: torch.Tensor, sens: torch.Tensor, xin: Act,
                x_out: torch.Tensor, key: str, m_next: Optional[torch.Tensor], dk_out: Optional[torch.Tensor] = None,
                stats_in: Optional[torch.Tensor] = None, stats_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same cascade on the state x = ifft2(k): the mask depends on kx only, so the soft data consistency is
        row-local there (ops.dc_rows).  xin channels 0, 1 already hold m = sum_c conj(S_c) x_c (written by the previous
        cascade's launch); this one writes x_out = x - w D(x) - r S and the next cascade's m into ``m_next``.
        stats_in: the statistics records of xin's planes the previous cascade's launch emitted; stats_out: where this one's
        launch puts those of ``m_next`` (ops.dc_rows_stat_part; the same buffer may serve both: it is read before it is written)."""
        n, c, h, w = x.shape
        r = ARENA.get(f"{key}.r", (n, 2, h, w), x.device)
        self.model.run(xin, r, key, part=stats_in)
        ops.dc_rows(x, sens, k0x, mask_f, self.dc_weight, r, x_out, m_next, dk_out,
                    m_stats=stats_out if m_next is not None else None)
        self._tape = (x, None, mask_f, sens, r, key)
        self._dk = dk_out
        return x_out

    def run_bwd_img(self, g_xout: torch.Tensor, g_sens: Optional[torch.Tensor], want_ref_grad: bool,
                    g_ref_acc: Optional[torch.Tensor] = None):
        """Backward of the last run_img().  g_xout = dL/dx' (complex; the caller may not reuse it).  Returns
        (dL/dx, dL/d ref or None).  With D_lin = ifft_x M fft_x (Hermitian):
            dL/dr = -sum_c conj(S_c) g_c          dL/dw = -Re sum conj(fft_x g) M (fft_x x - k0x)
            dL/dx = g - w D_lin(g) + g_m S        (g_m = NormUnet backward of dL/dr, through m = sum_c conj(S_c) x_c)"""
        x, _, mask_f, sens, r, key = self._tape
        n, c, h, w = x.shape
        dev = x.device
        g_r = ARENA.get("bwd.g_r", (n, 2, h, w), dev)
        g_d = torch.empty_like(g_xout)
        # dc_weight's gradient partials, the regulariser-input gradient's way into g_d and the sensitivity-map accumulation
        # ride in the LAST launch of the regulariser's backward (NormUnet.run_bwd, fuse=)
        part = ops.dc_rows_bwd(g_xout, sens, mask_f, self.dc_weight, g_d, g_r, self._dk, defer_dcw=True)
        _, g_ref = self.model.run_bwd(g_r, key, want_ref_grad, g_ref_acc,
                                      fuse=dict(gd=g_d, sens=sens, gS=g_sens, r=r, t1=g_xout, xs=x, dcw=(part, _grad_of(self.dc_weight))))
        return g_d, g_ref

    def forward(self, current_kspace: torch.Tensor, ref_kspace: torch.Tensor, mask: torch.Tensor,
                sens_maps: torch.Tensor, ref_image: Optional[torch.Tensor]) -> torch.Tensor:
        n, c, h, w = current_kspace.shape
        dev = current_kspace.device
        xin = self.model.input_buffer(n, h, w, dev, "cas")
        if self.model.use_ref:
            self.model.set_ref(xin, ref_image.contiguous())
        mask_f = mask.reshape(-1).to(torch.float32).contiguous()
        k_out = torch.empty_like(current_kspace)
        return self.run(current_kspace.contiguous(), ref_kspace.contiguous(), mask_f, sens_maps.contiguous(), xin,
                        k_out, "cas")


class VarNet(nn.Module):
    """End-to-end variational network.  Reference: varnet.py:422-486."""

    def __init__(self, num_cascades: int = 12, sens_chans: int = 8, sens_pools: int = 4, chans: int = 18,
                 pools: int = 4, mask_center: bool = True, use_ref: bool = False):
        super().__init__()
        self.use_ref = use_ref
        self.sens_net = SensitivityModel(chans=sens_chans, num_pools=sens_pools, mask_center=mask_center)
        self.cascades = nn.ModuleList([VarNetBlock(NormUnet(chans, pools, use_ref=use_ref)) for _ in range(num_cascades)])

    def forward(self, masked_kspace: torch.Tensor, mask: torch.Tensor, ref: Optional[torch.Tensor],
                num_low_frequencies: int) -> torch.Tensor:
        """varnet.py:465-486.  With autograd recording and parameters that require gradients the result carries a
        ``grad_fn`` (autograd._VarNetFn) whose backward is ``VarNet.backward``: ``ssimloss(net(...), target).backward()``
        fills every ``p.grad`` (varnet.py:559-560)."""
        from . import autograd
        with ops.use_arena(ops.owner_arena(self), outer_only=True):
            return autograd.varnet_forward(self, masked_kspace, mask, ref, num_low_frequencies)

    def _forward_impl(self, masked_kspace: torch.Tensor, mask: torch.Tensor, ref: Optional[torch.Tensor],
                      num_low_frequencies: int, retain: bool) -> torch.Tensor:
        """retain: keep every cascade's activations, state and data-consistency residual for backward()."""
        masked_kspace = masked_kspace.detach().contiguous()
        n, c, h, w = masked_kspace.shape
        dev = masked_kspace.device
        self._fwd_id = getattr(self, "_fwd_id", 0) + 1
        pre = self.__dict__.pop("_sens_pre", None)
        join_sens = None
        self._sens_fwd_arena = None
        if pre is not None and pre[1] == (masked_kspace.data_ptr(), tuple(masked_kspace.shape), num_low_frequencies):
            # CSModel issued the sensitivity network beside the alignment network (model.py: _sens_fork): its maps (and its
            # tapes, in its own arena) exist already; the main stream joins that stream in front of the first reader
            sens, aux_stream = pre[0], pre[2]
            self._sens_fwd_arena = pre[3]
            join_sens = lambda: ops._lib.rec(torch.cuda.current_stream().wait_stream, aux_stream)      # noqa: E731
        else:
            sens = self.sens_net(masked_kspace, num_low_frequencies)
        with ops._lib.untracked():                       # (a constant of the model: the sampling mask as floats)
            mask_f = mask.reshape(-1).to(torch.float32).contiguous()
        assert mask_f.numel() == w, "mask must be a [W] column mask (broadcast like the reference's [1,1,1,W])"
        ref1 = None
        if self.use_ref:
            ref = ref.detach().contiguous()
            ref1 = ops.rss(ref)                          # varnet.py:475-476
        # The cascades run on the IMAGE-domain state x_j = ifft2(k_j) (VarNetBlock.run_img): k0x = ifft_y(k0) is the data
        # term, x_0 = ifft2(k0), m_0 = sum_c conj(S_c) x_0; every cascade is then ONE row-local launch besides its U-Net,
        # and the output rss(ifft2(k_T)) (varnet.py:484-486) is rss(x_T).
        k0x = ARENA.get("cas.k0x", (n, c, h, w), dev, dtype=torch.complex64)
        ops.fft_cols(masked_kspace, True, out=k0x)
        T = len(self.cascades)
        if not retain:
            xin = self.cascades[0].model.input_buffer(n, h, w, dev, "cas") if T else None
            x = ARENA.get("cas.x", (n, c, h, w), dev, dtype=torch.complex64)
            ops.fft2c(masked_kspace, inverse=True, out=x)
            if join_sens is not None:
                join_sens()
            if T:
                if self.use_ref:
                    self.cascades[0].model.set_ref(xin, ref1)
                ops.sens_reduce(masked_kspace, sens, xin.buf)
            st = ops.dc_rows_stat_part(n, c, h, w, dev, ARENA) if (DC_STATS[0] and T > 1) else None
            for j, cascade in enumerate(self.cascades):
                cascade.run_img(x, k0x, mask_f, sens, xin, x, "cas", xin.buf if j + 1 < T else None,
                                stats_in=st if j > 0 else None, stats_out=st)
            return ops.rss(x)
        # training: every cascade keeps its own activations, state and data-consistency residual
        x = ARENA.get("cas.x0", (n, c, h, w), dev, dtype=torch.complex64)
        ops.fft2c(masked_kspace, inverse=True, out=x)
        xins = [cascade.model.input_buffer(n, h, w, dev, f"cas{j}") for j, cascade in enumerate(self.cascades)]
        if self.use_ref and T:
            # every cascade reads the SAME InstanceNorm-ed reference as channel 2 of its own input buffer (varnet.py:315-319):
            # normalised once, then copied (plane + affine entries) into the other buffers by one launch
            self.cascades[0].model.set_ref(xins[0], ref1)
            if T > 1:
                ops.replicate_channel(xins[0], xins[1:], 2)
        if join_sens is not None:
            join_sens()
        if T:
            ops.sens_reduce(masked_kspace, sens, xins[0].buf)
        st = ops.dc_rows_stat_part(n, c, h, w, dev, ARENA) if (DC_STATS[0] and T > 1) else None
        for j, cascade in enumerate(self.cascades):
            key = f"cas{j}"
            x_out = ARENA.get(f"{key}.xout", (n, c, h, w), dev, dtype=torch.complex64)
            dk = ARENA.get(f"{key}.dk", (n, c, h, w), dev, dtype=torch.complex64)
            x = cascade.run_img(x, k0x, mask_f, sens, xins[j], x_out, key, xins[j + 1].buf if j + 1 < T else None, dk,
                                stats_in=st if j > 0 else None, stats_out=st)
        out = ops.rss(x)
        self._train_state = (x, out.detach(), ref, ref1)      # (a detached alias: the returned tensor will carry the grad_fn)
        return out

    def backward(self, g_img: torch.Tensor, want_ref_grad: bool = False) -> Optional[torch.Tensor]:
        """Backward of the last training-mode forward().  g_img = dL/d(output image) [N,1,H,W].
        Accumulates every parameter gradient (cascades, dc weights, sensitivity net) and returns
        dL/d(ref) (the `ref` argument of forward) when asked."""
        with ops.use_arena(ops.owner_arena(self), outer_only=True), ops.backward_scope(g_img.device):
            return self._backward_impl(g_img, want_ref_grad)

    def _backward_impl(self, g_img: torch.Tensor, want_ref_grad: bool) -> Optional[torch.Tensor]:
        x_last, out, ref, ref1 = self._train_state
        g_x = ops.rss_bwd(x_last, out, g_img.contiguous())          # dL/dx_T: the state is already in the image domain
        g_sens = torch.empty_like(x_last)
        ops._lib.rec(g_sens.zero_)
        g_ref1 = None
        hook = getattr(self, "_grad_hook", None)         # data-parallel training: called as each part's gradients become final
        for j in reversed(range(len(self.cascades))):
            # (the first cascade visited creates dL/d ref, the others add to it inside their activation-backward kernel)
            g_x, g_ref = self.cascades[j].run_bwd_img(g_x, g_sens, want_ref_grad and self.use_ref, g_ref1)
            if g_ref is not None:
                g_ref1 = g_ref
            if hook is not None:
                hook(j)                                  # cascade j's weight and dc_weight gradients are complete
        # x_0 = ifft2(k0) and m_0 depend on the sensitivity maps only through m_0 = sum_c conj(S_c) x_0, which
        # run_bwd_img of cascade 0 has already accounted for; k0 itself needs no gradient
        arena = getattr(self, "_sens_fwd_arena", None)   # the sensitivity net's forward ran in its own arena (CSModel._sens_fork)
        aux = self.__dict__.get("_sens_async")           # a stream: CSModel runs this branch beside the alignment net's backward
        if arena is not None and aux is not None:
            ops._lib.rec(aux.wait_stream, torch.cuda.current_stream())         # dL/d(sens maps) is complete on the main stream
            g_sens.record_stream(aux)
            with ops.aux_region(aux, arena):
                self.sens_net.backward(g_sens)
            self._sens_bwd_stream = aux                  # whoever reads the sensitivity net's gradients waits for it
        elif arena is not None:
            with ops.use_arena(arena):
                self.sens_net.backward(g_sens)
        else:
            self.sens_net.backward(g_sens)
        if hook is not None:
            hook("sens")
        if g_ref1 is None:
            return None
        return ops.rss_bwd(ref, ref1, g_ref1)            # through ref = rss(ref)

