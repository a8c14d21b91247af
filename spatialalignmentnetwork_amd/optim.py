"""AdamW for the reconstruction / alignment networks (reference: torch.optim.AdamW(params, lr=1e-4,
weight_decay=0), model.py:72-87) as ONE fused HIP launch per network over flat buffers.

``FusedAdamW`` is a ``torch.optim.Optimizer`` (``param_groups`` / ``lr`` behave as usual, BaseModel's
isinstance checks keep working); the state lives in a ``dist.ParamBucket`` built lazily on the first
``bucket()`` call once the parameters are on the GPU.  There is no CPU path."""
from __future__ import annotations

import torch

from . import ops
from .dist import ParamBucket


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdamW keeps one flat buffer: a single parameter group")
        self._bucket = None
        self.device_step = False      # True: keep the step count on the device (set it BEFORE capturing update() into a hipGraph)
        self._hyper = None            # device fp32 [3] = (lr, weight_decay, grad_scale) of the graph-safe form
        self._hyper_host = None

    def _params(self):
        return self.param_groups[0]["params"]

    def bucket(self) -> ParamBucket:
        """The flat (param, grad, exp_avg, exp_avg_sq) buffers; rebuilt if the parameters moved."""
        if self._bucket is None or not self._bucket.owns(self._params()):
            old = self._bucket
            self._bucket = ParamBucket(self._params())
            if old is not None and old.total == self._bucket.total:          # e.g. after module.to(device)
                dev = self._bucket.flat.device
                self._bucket.exp_avg.copy_(old.exp_avg.to(dev))
                self._bucket.exp_avg_sq.copy_(old.exp_avg_sq.to(dev))
                self._bucket.steps = old.steps
        return self._bucket

    def zero_grad(self, set_to_none: bool = False):
        self.bucket().zero()

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        b = self.bucket()
        g = self.param_groups[0]
        if self.device_step or torch.cuda.is_current_stream_capturing():
            # graph-safe form: the step count lives in device memory (a captured launch cannot change its arguments)
            if b.step_dev is None:
                b.step_dev = torch.tensor([b.steps], dtype=torch.int64, device=b.flat_p.device)
            self.device_step = True
            # lr / weight decay / gradient scale live in device memory too, so that a captured launch follows
            # param_groups['lr'] (sync_hyper() refreshes them; an eager step calls it itself)
            if not torch.cuda.is_current_stream_capturing():
                from . import _lib
                with _lib.untracked():          # (a 12-byte host-to-device copy when a value changed; never part of a replay)
                    self.sync_hyper(grad_scale)
            elif self._hyper is None:
                raise RuntimeError("FusedAdamW: run one eager step (or sync_hyper()) before capturing")
            ops.adamw_step_dev(b.flat_p, b.flat, b.exp_avg, b.exp_avg_sq, float(g["lr"]), float(g["betas"][0]),
                               float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), b.step_dev, float(grad_scale),
                               self._hyper)
        else:
            b.steps += 1
            ops.adamw_step(b.flat_p, b.flat, b.exp_avg, b.exp_avg_sq, float(g["lr"]), float(g["betas"][0]),
                           float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), b.steps, float(grad_scale))
        ops.bump_weight_epoch()

    def sync_hyper(self, grad_scale: float = None) -> None:
        """Copy (lr, weight_decay, grad_scale) to the device if they changed (a stream-ordered 12-byte copy, never inside
        a capture).  Call it after changing ``param_groups[0]['lr']`` between replays of a captured step."""
        g = self.param_groups[0]
        gs = self._hyper_host[2] if (grad_scale is None and self._hyper_host is not None) else float(1.0 if grad_scale is None else grad_scale)
        want = (float(g["lr"]), float(g["weight_decay"]), gs)
        if self._hyper is None or self._hyper.device != self.bucket().flat_p.device:
            self._hyper = torch.empty(3, dtype=torch.float32, device=self.bucket().flat_p.device)
            self._hyper_host = None
        if self._hyper_host != want:
            self._hyper.copy_(torch.tensor(want, dtype=torch.float32), non_blocking=False)
            self._hyper_host = want

    def steps_taken(self) -> int:
        """Optimisation steps so far (reads the device counter in the graph-safe form: a host synchronisation)."""
        b = self.bucket()
        return int(b.step_dev.item()) if (self.device_step and b.step_dev is not None) else b.steps
