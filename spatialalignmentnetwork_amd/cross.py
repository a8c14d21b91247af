"""Host-side mirror of the reference's cross.py (SpatialTransformer) on the HIP
kernels.  Reference: cross.py:9-38."""
from __future__ import annotations

import torch

from . import ops
from .ops import Act, GLOBAL_ARENA as ARENA
from .unet import UNet


class SpatialTransformer(torch.nn.Module):
    """Alignment network: UNet(2C -> 32) -> LeakyReLU -> conv3x3(32 -> 2) predicts a
    displacement field in normalised coordinates (x, y); ``warp`` resamples with it."""

    def __init__(self, channels=1):
        super().__init__()
        self.channels = channels
        self.net = torch.nn.Sequential(
            UNet(2 * channels, 32, (32, 64, 64, 64, 64)),
            torch.nn.LeakyReLU(inplace=True),
            torch.nn.Conv2d(32, 2, kernel_size=3, padding=1))
        # the reference zero-initialises the head so training starts from the identity warp (cross.py:20-21)
        torch.nn.init.zeros_(self.net[-1].weight)
        torch.nn.init.zeros_(self.net[-1].bias)

    def forward(self, moving, fixed, features=None):
        """Returns (offset [N,H,W,2], grid [N,H,W,2]) like cross.py:23-30.  The
        offset is a permuted view of the head's NCHW output, as in the reference.  With autograd recording both carry
        a ``grad_fn`` (autograd._AlignFn) whose backward is ``SpatialTransformer.backward``."""
        from . import autograd
        with ops.use_arena(ops.owner_arena(self), outer_only=True):
            offset_nchw, grid = autograd.align_forward(self, moving, fixed)
        return offset_nchw.permute(0, 2, 3, 1), grid

    def _forward_impl(self, moving, fixed, retain: bool):
        moving, fixed = moving.detach(), fixed.detach()
        self._fwd_id = getattr(self, "_fwd_id", 0) + 1
        n, c, h, w = moving.shape
        dev = moving.device
        xin = Act(ARENA.get("align.in", (n, 2 * c, h, w), dev), 0, 2 * c)
        ops.apply(ops.full(moving.contiguous()), xin.view(0, c))      # torch.cat([moving, fixed], 1)
        ops.apply(ops.full(fixed.contiguous()), xin.view(c, c))
        feat = Act(ARENA.get("align.feat", (n, 32, h, w), dev), 0, 32, None, None, 0.01)   # LeakyReLU read lazily
        self.net[0].run(xin, feat, retain=retain)
        head = self.net[2]
        offset_nchw = torch.empty((n, 2, h, w), device=dev)
        ops.conv2d(feat, head.weight, head.bias, ops.full(offset_nchw))
        grid = torch.empty((n, h, w, 2), device=dev)
        # grid = affine_grid(identity) + offset (cross.py:24-29): c == 0 asks the warp kernel for the grid only
        ops.lib().call("san_warp_fwd", ops._p(None), ops._p(offset_nchw), ops._p(None), ops._p(grid), n, 0, h, w, 0,
                       ops._stream())
        self._last_offset_nchw = offset_nchw.detach()
        self._feat = feat
        return offset_nchw, grid

    def backward(self, g_offset_nchw: torch.Tensor) -> None:
        """dL/d(offset) (NCHW [N,2,H,W]: the sum of the warp's grid gradient and the smoothness
        term) -> gradients of every alignment-network parameter (training-mode forward required)."""
        with ops.use_arena(ops.owner_arena(self), outer_only=True), ops.backward_scope(g_offset_nchw.device):
            self._backward_impl(g_offset_nchw)

    def _backward_impl(self, g_offset_nchw: torch.Tensor) -> None:
        from .unet import _grad_of
        head, feat = self.net[2], self._feat
        dy = ops.full(g_offset_nchw.contiguous())
        part = ops.plane_stats(dy, tag="st.b")
        gb = _grad_of(head.bias)
        ops._lib.rec(lambda: gb.add_((part[..., 0] * part[..., 1]).sum(dim=(0, 2))))
        ops.conv2d_wgrad(feat, dy, _grad_of(head.weight), accumulate=True)
        g_feat = Act(ARENA.get("align.g_feat", tuple(feat.buf.shape), feat.buf.device), 0, feat.c)
        ops.conv2d_dgrad(dy, head.weight, g_feat)
        self.net[0].run_bwd(feat, g_feat)

    def warp(self, img, grid, interp=False):
        """Bilinear, zeros padding, align_corners=False; inputs forced to fp32 (cross.py:32-38).  Differentiable wrt the
        grid (and, with float atomics, the image) through autograd._WarpFn."""
        from . import autograd
        warped = autograd.warp(img.float(), grid.float())
        if interp and warped.shape != img.shape:
            # cross.py:35-37: F.interpolate(warped, size=img.shape[2:]) in its default mode 'nearest' --
            # source index = min(floor(dst * in / out), in - 1) with the ratio taken in float32, as ATen computes it.
            # Two index_selects (differentiable); not on CSModel's path (model.py never passes interp=True).
            for dim, (n_in, n_out) in ((2, (warped.shape[2], img.shape[2])), (3, (warped.shape[3], img.shape[3]))):
                if n_in != n_out:
                    scale = torch.tensor(n_in / n_out, dtype=torch.float32)
                    idx = torch.clamp((torch.arange(n_out, dtype=torch.float32) * scale).floor().long(), max=n_in - 1)
                    warped = warped.index_select(dim, idx.to(warped.device))
        return warped
