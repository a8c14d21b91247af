"""torch.autograd.Function wrappers over the hand-written backward chain.

The reference's seam is ``nn.Module.forward`` + ``loss.backward()`` (varnet.py:559-560, model.py:203-214).  The
functions below make the module-level entry points of this package visible to autograd, so that idiom works:

    result = varnet(masked_kspace, mask, ref, nlf);  ssimloss(result, target).backward()

Every ``backward`` here calls the same HIP kernels ``CSModel.update()`` drives directly (``VarNet.backward``,
``SpatialTransformer.backward``, ``ops.*_bwd``), so both routes produce the same bits.

Parameter gradients are a SIDE EFFECT of the module Functions: ``VarNet.backward`` / ``SpatialTransformer.backward``
accumulate into ``p.grad`` (views of the optimiser's flat gradient buffer) while they walk the cascade; the Functions
take the parameters as inputs only so that autograd knows the output depends on them, and return ``None`` for them.
``loss.backward()`` therefore fills ``p.grad`` exactly like the reference; ``torch.autograd.grad(loss, params)`` does not
see those gradients (use ``p.grad``).  Each module keeps ONE tape: backward must follow the forward it belongs to.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from . import ops


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


class _Scope:
    """What a backward needs from the forward's surroundings: the arena that holds the tape and the conv arithmetic
    (autograd runs backward on its own thread, outside the ``with`` blocks of the forward)."""

    def __init__(self):
        self.arena = ops._ARENA_STACK[-1]
        self.mode = ops.current_precision()

    def __enter__(self):
        self._a = ops.use_arena(self.arena)
        self._p = ops.conv_precision(self.mode)
        self._a.__enter__()
        self._p.__enter__()
        return self

    def __exit__(self, *exc):
        self._p.__exit__(*exc)
        self._a.__exit__(*exc)
        return False


# ------------------------------------------------------------------------------------------------ signal_utils
class _Fft2Fn(Function):
    """ortho fft2 / ifft2 are unitary: the adjoint of one is the other (signal_utils.py:4-12)."""

    @staticmethod
    def forward(ctx, x, inverse):
        ctx.inverse = inverse
        return ops.fft2c(_c(x), inverse=inverse)

    @staticmethod
    def backward(ctx, g):
        return ops.fft2c(_c(g), inverse=not ctx.inverse), None


def fft2c(x: torch.Tensor, inverse: bool) -> torch.Tensor:
    if _needs_grad(x):
        return _Fft2Fn.apply(x, inverse)
    return ops.fft2c(_c(x), inverse=inverse)


class _RssFn(Function):
    """y = sqrt(sum_c |x_c|^2), gx_c = g x_c / y (signal_utils.py:24-26)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y = ops.rss(x)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        return ops.rss_bwd(x, y, _c(g))


def rss(x: torch.Tensor) -> torch.Tensor:
    if _needs_grad(x):
        return _RssFn.apply(x)
    return ops.rss(_c(x))


# ------------------------------------------------------------------------------------------------ losses
class _SsimLossFn(Function):
    """ssimloss.py:11-40; SSIM is symmetric in (X, Y), so one backward kernel serves both arguments."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _c(x), _c(y)
        ctx.save_for_backward(x, y)
        return ops.ssim_loss(x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = ops.ssim_loss_bwd(y, x, 1.0, gdev=g) if ctx.needs_input_grad[0] else None
        gy = ops.ssim_loss_bwd(x, y, 1.0, gdev=g) if ctx.needs_input_grad[1] else None
        return gx, gy


def ssimloss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    if _needs_grad(x, y):
        return _SsimLossFn.apply(x, y)
    return ops.ssim_loss(_c(x), _c(y))


class _LnccLossFn(Function):
    """lnccloss.py:7-56."""

    @staticmethod
    def forward(ctx, i, j, win):
        i, j = _c(i), _c(j)
        ctx.save_for_backward(i, j)
        ctx.win = win
        return ops.lncc_loss(i, j, win)

    @staticmethod
    def backward(ctx, g):
        i, j = ctx.saved_tensors
        gi, gj = ops.lncc_loss_bwd(i, j, ctx.needs_input_grad[0], ctx.needs_input_grad[1], 1.0, gdev=g, win=ctx.win)
        return gi, gj, None


def lncc_loss(i: torch.Tensor, j: torch.Tensor, win: int = 9) -> torch.Tensor:
    if _needs_grad(i, j):
        return _LnccLossFn.apply(i, j, win)
    return ops.lncc_loss(_c(i), _c(j), win)


class _SmoothPoolFn(Function):
    """avg_pool2(gaussian_smooth(x)) between the scales of ms_lncc_loss (lnccloss.py:59-60, miloss.py:20-24)."""

    @staticmethod
    def forward(ctx, x, kern):
        ctx.save_for_backward(kern)
        ctx.in_hw = (int(x.shape[2]), int(x.shape[3]))      # odd sizes: avg_pool2d drops the last row / column
        return ops.smooth_pool(_c(x), kern)

    @staticmethod
    def backward(ctx, g):
        (kern,) = ctx.saved_tensors
        return ops.smooth_pool_bwd(_c(g), kern, in_hw=ctx.in_hw), None


def smooth_pool(x: torch.Tensor, kern: torch.Tensor) -> torch.Tensor:
    if _needs_grad(x):
        return _SmoothPoolFn.apply(x, kern)
    return ops.smooth_pool(_c(x), kern)


class _GradientLossFn(Function):
    """model.py:21-28 on the NCHW storage of the offset field."""

    @staticmethod
    def forward(ctx, off_nchw):
        off_nchw = _c(off_nchw)
        ctx.save_for_backward(off_nchw)
        return ops.gradient_loss_nchw(off_nchw)

    @staticmethod
    def backward(ctx, g):
        (off,) = ctx.saved_tensors
        out = torch.empty_like(off)
        ops.gradient_loss_bwd(off, out, 1.0, False, gdev=g)
        return out


def gradient_loss_nchw(off_nchw: torch.Tensor) -> torch.Tensor:
    if _needs_grad(off_nchw):
        return _GradientLossFn.apply(off_nchw)
    return ops.gradient_loss_nchw(_c(off_nchw))


# ------------------------------------------------------------------------------------------------ warp
class _WarpFn(Function):
    """F.grid_sample(img, grid, bilinear, zeros, align_corners=False) (cross.py:32-34).  The grid gradient comes out of
    the kernel in NCHW order and is handed to autograd as the NHWC-shaped permuted view (no copy)."""

    @staticmethod
    def forward(ctx, img, grid):
        img, grid = _c(img), _c(grid)
        ctx.save_for_backward(img, grid)
        return ops.grid_sample(img, grid)

    @staticmethod
    def backward(ctx, g):
        img, grid = ctx.saved_tensors
        g = _c(g)
        g_img = g_grid = None
        if ctx.needs_input_grad[1]:
            if tuple(grid.shape[1:3]) != tuple(img.shape[2:]):
                raise NotImplementedError("grid gradient of a resampling warp (output size != image size)")
            g_grid = ops.warp_bwd_grid(img, grid, g).permute(0, 2, 3, 1)
        if ctx.needs_input_grad[0]:
            g_img = ops.grid_sample_bwd_img(grid, g, tuple(img.shape))
        return g_img, g_grid


def warp(img: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    if _needs_grad(img, grid):
        return _WarpFn.apply(img, grid)
    return ops.grid_sample(_c(img), _c(grid))


# ------------------------------------------------------------------------------------------------ modules
def _stale(name):
    return RuntimeError(f"{name}: backward through a forward that is no longer the module's latest one (each module keeps "
                        "one tape: run backward before the next training-mode forward)")


class _VarNetFn(Function):
    """VarNet.forward (varnet.py:465-486) with VarNet.backward as its adjoint.  Inputs after ``nlf`` are the module's
    parameters (dependency markers: their gradients are accumulated into p.grad by the backward chain itself)."""

    @staticmethod
    def forward(ctx, net, masked_kspace, mask, ref, nlf, *params):
        out = net._forward_impl(masked_kspace, mask, ref, nlf, retain=True)
        ctx.net, ctx.fwd_id, ctx.scope, ctx.nparams = net, net._fwd_id, _Scope(), len(params)
        return out

    @staticmethod
    def backward(ctx, g):
        net = ctx.net
        if net._fwd_id != ctx.fwd_id:
            raise _stale("VarNet")
        want_ref = ctx.needs_input_grad[3]
        with ctx.scope, ops.backward_scope(g.device):
            g_ref = net.backward(_c(g), want_ref_grad=want_ref)
        return (None, None, None, g_ref if want_ref else None, None) + (None,) * ctx.nparams


# Bumped whenever ANY module registers a parameter or a submodule (torch's global registration hooks): the memoised parameter
# lists below are rebuilt then, so a replaced Parameter or a swapped submodule is noticed without walking the module tree on
# every forward.
_PARAM_EPOCH = [0]


def _bump_param_epoch(*_a, **_k):
    _PARAM_EPOCH[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump_param_epoch)
torch.nn.modules.module.register_module_module_registration_hook(_bump_param_epoch)


def _trainable(mod):
    """The module's parameters that require gradients (memoised: walking 300-odd parameters per forward was 2 ms of host
    time per training step).  The parameter list is rebuilt when a parameter / submodule was registered anywhere since
    (_PARAM_EPOCH), the selection when a requires_grad flag changed."""
    hit = mod.__dict__.get("_san_trainable")
    plist = mod.__dict__.get("_san_plist")
    if plist is None or plist[0] != _PARAM_EPOCH[0]:
        plist = (_PARAM_EPOCH[0], list(mod.parameters()))
        object.__setattr__(mod, "_san_plist", plist)
        hit = None
    plist = plist[1]
    flags = tuple(p.requires_grad for p in plist)
    if hit is None or hit[0] != flags:
        hit = (flags, tuple(p for p in plist if p.requires_grad))
        object.__setattr__(mod, "_san_trainable", hit)
    return hit[1]


def varnet_forward(net, masked_kspace, mask, ref, nlf):
    params = _trainable(net)
    if torch.is_grad_enabled() and (params or _needs_grad(ref)):
        if _needs_grad(masked_kspace):
            raise NotImplementedError("VarNet: no gradient path wrt the measured k-space (it is data)")
        return _VarNetFn.apply(net, masked_kspace, mask, ref, nlf, *params)
    return net._forward_impl(masked_kspace, mask, ref, nlf, retain=False)


class _AlignFn(Function):
    """SpatialTransformer.forward (cross.py:23-30): returns (offset NCHW, grid NHWC); grid = identity + offset, so the
    two incoming gradients add up to dL/d(offset)."""

    @staticmethod
    def forward(ctx, st, moving, fixed, *params):
        off, grid = st._forward_impl(moving, fixed, retain=True)
        ctx.st, ctx.fwd_id, ctx.scope, ctx.nparams = st, st._fwd_id, _Scope(), len(params)
        return off, grid

    @staticmethod
    def backward(ctx, g_off, g_grid):
        st = ctx.st
        if st._fwd_id != ctx.fwd_id:
            raise _stale("SpatialTransformer")
        parts = []
        if g_off is not None:
            parts.append(_c(g_off))
        if g_grid is not None:
            parts.append(_c(g_grid.permute(0, 3, 1, 2)))
        if parts:
            total = parts[0]
            if len(parts) == 2:
                total = torch.empty_like(parts[0])
                ops.add(ops.full(parts[0]), ops.full(parts[1]), ops.full(total))
            with ctx.scope, ops.backward_scope(total.device):
                st.backward(total)
        return (None, None, None) + (None,) * ctx.nparams


def align_forward(st, moving, fixed):
    params = _trainable(st)
    if torch.is_grad_enabled() and params:
        if _needs_grad(moving, fixed):
            raise NotImplementedError("SpatialTransformer: no gradient path wrt the input images (they are data)")
        return _AlignFn.apply(st, moving, fixed, *params)
    return st._forward_impl(moving, fixed, retain=False)
