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
            ops.adamw_step_dev(b.flat_p, b.flat, b.exp_avg, b.exp_avg_sq, float(g["lr"]), float(g["betas"][0]),
                               float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), b.step_dev, float(grad_scale))
        else:
            b.steps += 1
            ops.adamw_step(b.flat_p, b.flat, b.exp_avg, b.exp_avg_sq, float(g["lr"]), float(g["betas"][0]),
                           float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), b.steps, float(grad_scale))
        ops.bump_weight_epoch()

    def steps_taken(self) -> int:
        """Optimisation steps so far (reads the device counter in the graph-safe form: a host synchronisation)."""
        b = self.bucket()
        return int(b.step_dev.item()) if (self.device_step and b.step_dev is not None) else b.steps
