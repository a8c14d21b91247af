"""Host-side mirror of the reference's CSModel (model.py:39-321) for the
reconstruction + alignment path: same constructor, ``set_input`` / ``forwardT`` /
``forwardR`` / ``test`` / ``get_vis`` / ``save`` / ``load`` protocol and the same
``img_*`` / ``loss_*`` / ``metric_*`` attribute discovery, driving the HIP kernels.

Scope notes: the GAN branch (``net_G`` / ``net_D``, regimes
'Mixed' and 'GAN-Only') is out of scope for this path, so ``test()`` does not run
``forwardG`` and reports no ``loss_gan_sim``; ``num_cascades`` etc. are
constructor-visible through cfg (defaults = the reference's hard-coded values,
model.py:64-71) instead of being hard-coded.
"""
from __future__ import annotations

import ctypes
import math
import os

import torch

from . import _lib, autograd, ops
from .basemodel import BaseModel, Config  # noqa: F401
from .cross import SpatialTransformer
from . import metrics
from .masks import masks
from .optim import FusedAdamW
from .signal_utils import rss
from .ssimloss import ssimloss
from .varnet import VarNet


def gradient_loss(s: torch.Tensor) -> torch.Tensor:
    """Smoothness of an NHWC offset field (model.py:21-28).  ``s`` must be the
    permuted view SpatialTransformer.forward returns (NCHW storage)."""
    assert s.shape[-1] == 2, "not 2D grid?"
    return autograd.gradient_loss_nchw(s.permute(0, 3, 1, 2))     # differentiable (autograd._GradientLossFn)


# CSModel.update() records the step (record_update) once it has seen AUTO_AFTER identical eager steps and replays it from then
# on, so that the reference's own training loop (train.py:212-217: ``net.set_input(*batch); net.update()``) runs at the
# recorded step's speed.  SAN_AUTO_RECORD=0 or cfg.auto_record = False: always eager.
AUTO_RECORD = [os.environ.get("SAN_AUTO_RECORD", "1") != "0"]
# Data-parallel gradient exchange of net_R: "cascade" (default) = one all-reduce per cascade in reverse order, launched from
# inside VarNet.backward as each cascade's gradients become final; "single" = the whole flat buffer afterwards.
AUTO_AFTER = 2
AUTO_KEEP = 3          # recordings kept besides the current one (each holds its step's tensors: ~6 GB at N = 8, 320^2)
# Sensitivity network on an auxiliary stream beside the alignment network (forward and backward).  On by default again in round 5:
# the misread that made co-resident kernels of two streams disagree needs a packed-fp32 instruction in the victim
# (scratch/probe/pk32_two_stream_repro.hip), the library has none (build.py NO_PK32, tests/test_abi.py), and
# tests/test_gpu_step_runtime.py holds 50 overlapped steps at N = 8, 320^2 and 10 at 15 x 640 x 368 to the serial step bit for bit.
SENS_OVERLAP_DEFAULT = os.environ.get("SAN_SENS_OVERLAP", "1") != "0"
SENS_OVERLAP = [SENS_OVERLAP_DEFAULT]
_SENS_DBG = int(os.environ.get("SAN_SENS_DBG", "0"))      # 1: forward branch only, 2: backward branch only (debugging)


def _no_auto(fn):
    """The explicit recorders / capturers run their warm-up and recorded steps eagerly (no nested auto-recording)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        prev = getattr(self, "_auto_busy", False)
        self._auto_busy = True
        try:
            return fn(self, *a, **k)
        finally:
            self._auto_busy = prev
    return wrapped


def _own_arena(fn):
    """Run a CSModel method with the model's own arena current (ops.use_arena): two models in one process never share
    activation tapes."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with ops.use_arena(ops.owner_arena(self)):
            return fn(self, *a, **k)
    return wrapped


class CSModel(BaseModel):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.memo_init = set(self.__dict__.keys()) | {"memo_init", "_aux_abs", "_replicas_synced", "conv_dtype", "bwd_dtype", "_san_arena", "_exchange",
                                                         "_exchange_events", "_exchange_wait_events", "exchange_slices", "_aux_stream", "_split_capture", "time_exchange", "_auto", "_auto_cache", "_auto_busy", "auto_record", "step_mode", "_input_pending", "_sens_mark"}

    def build(self, cfg):
        super().build(cfg)
        assert cfg.lr == 1e-4          # model.py:52
        shape, sparsity, coils = cfg.shape, cfg.sparsity, cfg.coils
        get = lambda k, d: cfg[k] if k in cfg else d
        mask = cfg.mask
        if mask not in masks:
            raise NotImplementedError(f"mask {mask!r}: only {sorted(masks)} are built (the learned LOUPE / Taylor masks "
                                      "of masks.py:71-84,127-166 are outside this path)")
        self.net_mask = masks[mask](shape) if mask == "mask" else masks[mask](sparsity, shape)
        self.net_T = SpatialTransformer(channels=coils)
        self.net_R = VarNet(num_cascades=get("num_cascades", 8), sens_chans=get("sens_chans", 8),
                            sens_pools=get("sens_pools", 4), chans=get("chans", 18), pools=get("pools", 4),
                            use_ref=True)
        self.optim_T = FusedAdamW(self.net_T.parameters(), lr=cfg.lr, weight_decay=0)
        self.optim_R = FusedAdamW(self.net_R.parameters(), lr=cfg.lr, weight_decay=0)
        self.use_amp = bool(get("use_amp", False))
        # The reference's mixed-precision seam is torch.cuda.amp.autocast(enabled=use_amp) around the forwards
        # (model.py:83-87,104).  Here it selects the arithmetic of the matrix-core convolutions: cfg.conv_dtype in
        # {"bf16x3" (fp32-equivalent, default), "bf16x2", "bf16", "fp8" (e4m3 forward, bf16 gradients)}; use_amp without conv_dtype means plain bf16.  FFT, data
        # consistency, normalisation statistics and losses are fp32 in every mode; no GradScaler is needed (bf16 has
        # fp32's exponent range).
        self.conv_dtype = get("conv_dtype", None)      # None: follow use_amp (which eval.py:41 switches off after loading)
        # cfg.bwd_dtype (optional): arithmetic of the BACKWARD convolutions (data and weight gradients) only, e.g. "bf16x2"
        # under an fp32-equivalent forward: the outputs keep the 1e-4 parity, the gradients carry 16 mantissa bits
        # (4e-6 per layer, against the 1e-2-level fp32-vs-fp64 noise of the reference's own gradients).  Default: as forward.
        self.bwd_dtype = get("bwd_dtype", None)
        if self.bwd_dtype is not None and self.bwd_dtype not in ops.CONV_PRECISIONS:
            raise ValueError(f"cfg.bwd_dtype {self.bwd_dtype!r}: choose from {sorted(ops.CONV_PRECISIONS)}")
        if self.conv_dtype is not None and self.conv_dtype not in ops.CONV_PRECISIONS:
            raise ValueError(f"cfg.conv_dtype {self.conv_dtype!r}: choose from {sorted(ops.CONV_PRECISIONS)}")
        self.device = torch.device("cpu")

    # ------------------------------------------------------------------ inputs
    @_own_arena
    def set_input(self, img_full, img_aux=None):
        """fft2 -> drop pruned columns -> ifft2 -> rss x3.  model.py:89-121."""
        for name in [k for k in self.__dict__ if k.startswith(("loss_", "img_", "metric_"))]:
            delattr(self, name)
        extra = set(self.__dict__.keys()) - self.memo_init
        assert len(extra) == 0, extra
        self.img_full = img_full.contiguous()
        self.img_aux = torch.zeros_like(img_full) if img_aux is None else img_aux.contiguous()
        # ONCE per step (round 6): when update() will replay a recorded step for exactly this configuration, the recording
        # contains this prologue -- computing it here as well ran 3 FFT + 3 rss launches and their host work twice per step.
        # The derived img_* attributes then come out of the replay; a read before that (test(), get_vis(), user code)
        # computes them on demand (__getattr__ -> _materialize_input).
        st = self._auto_state() if self.training else None
        if st is not None and st["step"] is not None:
            self._input_pending = True
            return
        self._set_input_compute()

    def _set_input_compute(self):
        pruned = self.net_mask.pruned
        keep = _keep_mask(pruned)
        self.img_k_full = ops.fft2c(self.img_full)
        self.img_k_sampled = ops.fft2c(self.img_full, colmask_out=keep)      # k_full * (1 - pruned)
        self.img_sampled = ops.fft2c(self.img_k_sampled, inverse=True)
        self.img_full_rss = rss(self.img_full)
        self.img_sampled_rss = rss(self.img_sampled)
        self.img_aux_rss = rss(self.img_aux)
        n, _, h, w = self.img_full.shape
        with _lib.untracked():                   # (a constant of the model, for visualisation only)
            vis = (1.0 - pruned.float()).roll(w // 2).view(1, 1, 1, w).expand(n, 1, h, w)   # fftshift2 of a column mask
        self.img_mask = vis

    def _materialize_input(self) -> None:
        """The deferred half of set_input, if it is still owed (see set_input)."""
        if self.__dict__.get("_input_pending"):
            self._input_pending = False
            with ops.use_arena(ops.owner_arena(self)):
                self._set_input_compute()

    def __getattr__(self, name):
        # (only reached when normal lookup fails: an img_* attribute read between a deferred set_input and the replay)
        if name.startswith("img_") and self.__dict__.get("_input_pending"):
            self._materialize_input()
            if name in self.__dict__:
                return self.__dict__[name]
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

    # ---------------------------------------------------------------- forwards
    def _sens_fork(self) -> None:
        """The sensitivity network (varnet.py:389-420) depends on the sampled k-space only, the alignment network on the two
        images only: the former is issued on an auxiliary stream with its own arena while the latter runs on the main stream
        (1.9 of the 4.4 ms in front of the first cascade); VarNet._forward_impl joins.  Backward likewise (VarNet._backward_impl:
        1.2 ms beside the alignment network's 4.3 ms).  SAN_SENS_OVERLAP=0: in line, as the reference orders them."""
        R = self.net_R
        mark = self.__dict__.pop("_sens_mark", None)
        if mark is None:
            return
        aux = self._aux_stream
        arena = ops.owner_arena(R.sens_net)
        main = torch.cuda.current_stream()
        mk = self.img_k_sampled.detach().contiguous()
        nlf = int(self.cfg.shape * self.cfg.sparsity * 0.32)
        _lib.rec(aux.wait_event, mark)                  # (everything the sensitivity network reads was queued before the mark)
        with ops.aux_region(aux, arena):
            sens = R.sens_net(mk, nlf)
        R._sens_pre = (sens, (mk.data_ptr(), tuple(mk.shape), nlf), aux, arena)
        if _SENS_DBG == 2:
            _lib.rec(main.wait_stream, aux)

    def _sens_mark_fork(self) -> None:
        """The POINT of the main stream the sensitivity branch forks from (an event; stale weight images are re-packed first).
        The branch's ~200 launches are queued LATER, by _sens_fork at the end of forwardT: queued here, in front of the alignment
        network's launches, they kept the host busy for 0.55 ms during which the main queue had nothing to run (round 6,
        profiles/r06_main_queue_gaps.txt) -- now the host feeds the main stream first and the auxiliary one while that runs."""
        R = self.net_R
        if not SENS_OVERLAP[0] or self.device.type != "cuda" or not hasattr(R, "sens_net") or not hasattr(self, "img_k_sampled"):
            return
        if self.__dict__.get("_aux_stream") is None:
            self._aux_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        ops.ensure_packs(self.device)                   # (on the main stream, before the fork point)
        ev = _lib.light(torch.cuda.Event())
        _lib.rec(ev.record, main)
        self._sens_mark = ev

    def _sens_join(self) -> None:
        """The main stream waits for the sensitivity network's backward (its gradients are read by the exchange / the optimiser)."""
        st = self.net_R.__dict__.pop("_sens_bwd_stream", None)
        if st is not None:
            _lib.rec(torch.cuda.current_stream().wait_stream, st)

    @_own_arena
    def forwardT(self):
        """model.py:142-155."""
        self._sens_mark_fork()
        aux_abs = ops.cabs(self.img_aux)
        self._aux_abs = aux_abs
        self.img_offset, self.img_grid = self.net_T(moving=aux_abs, fixed=ops.cabs(self.img_sampled))
        self.img_warped = self.net_T.warp(aux_abs, self.img_grid)
        self.img_warped_rss = rss(self.img_warped)
        self.loss_smooth = gradient_loss(self.img_offset)
        with _lib.untracked():                   # (host-visible scalar arithmetic: the direct backward does not read it)
            self.loss_all = self.loss_all + self.loss_smooth * self.cfg.weight_smooth
        self._sens_fork()                        # (the sensitivity branch's launches: queued behind the alignment network's, see _sens_mark_fork)

    @_own_arena
    def forwardR(self):
        """model.py:157-169."""
        with _lib.untracked():
            keep = torch.logical_not(self.net_mask.pruned)
        self.img_rec = self.net_R(
            masked_kspace=self.img_k_sampled,
            mask=keep,
            ref=self.img_warped,
            num_low_frequencies=int(self.cfg.shape * self.cfg.sparsity * 0.32))
        self.loss_sim = ssimloss(self.img_full_rss, self.img_rec)
        with _lib.untracked():
            self.loss_all = self.loss_all + self.loss_sim * self.cfg.weight_sim

    @_own_arena
    def backward(self, train_T: bool) -> None:
        """Hand-written backward of loss_all = weight_smooth*loss_smooth + weight_sim*loss_sim through
        forwardR (VarNet) and, when train_T, through the warp into forwardT (alignment network).
        Replaces ``scalar.scale(loss_all).backward()`` (model.py:203-214); ``loss_all.backward()`` itself works too
        (autograd.py) and gives the same bits."""
        with ops.backward_scope(self.device):   # gradient-maximum records zeroed, weight gradients on the side stream
            try:
                self._backward(train_T)
            finally:
                self._sens_join()

    def _backward(self, train_T: bool) -> None:
        g_rec = ops.ssim_loss_bwd(self.img_full_rss, self.img_rec, float(self.cfg.weight_sim))
        exch = getattr(self, "_exchange", None)
        per_cascade = exch is not None             # ONE bucket form: net_R's buffer goes out cascade by cascade
        if per_cascade:
            bucket = self.optim_R.bucket()
            ranges = self._cascade_ranges(bucket)

            def hook(which):
                # cascade `which` (or the sensitivity net) has its gradients: its deferred weight-gradient reductions are
                # flushed on the side stream and its slice of the flat buffer goes out on the communication stream
                if ranges.get(which) is not None:
                    ops.wgrad_flush()
                    exch.launch(bucket, after=(ops._WG["stream"], self.net_R.__dict__.get("_sens_bwd_stream")), rng=ranges[which])

            self.net_R._grad_hook = hook
        if SENS_OVERLAP[0] and self.__dict__.get("_aux_stream") is not None and _SENS_DBG != 1:
            self.net_R._sens_async = self._aux_stream    # (this caller joins it: _sens_join)
        try:
            g_warped = self.net_R.backward(g_rec, want_ref_grad=train_T)
        finally:
            self.net_R._grad_hook = None
            self.net_R.__dict__.pop("_sens_async", None)
        if not train_T:
            return
        off = self.net_T._last_offset_nchw
        # dL/d(offset): through warp -> rss is already folded by VarNet.backward (returns dL/d warped)
        g_off = ops.warp_bwd_grid(self._aux_abs, self.img_grid.detach(), g_warped)
        ops.gradient_loss_bwd(off, g_off, float(self.cfg.weight_smooth), True)
        self.net_T.backward(g_off)

    def update(self):
        """One optimisation step (model.py:193-216).  The first AUTO_AFTER calls with a given configuration run eagerly; the
        next one records the step (``record_update`` on private copies of the inputs; the recording itself does not advance
        the model) and from then on every call copies the current ``img_full`` / ``img_aux`` into the recording's static inputs
        and replays it -- bit-identical to the eager step (tests/test_gpu_step_runtime.py).  Anything the recording depends on
        (shapes, regime, loss weights, convolution precision, world size, train / eval flags, requires_grad pattern, parameter
        and mask storage) is part of a key that is compared on every call: a change drops the recording and the step runs
        eagerly again.  ``step_mode`` says which form the last call took."""
        st = self._auto_state()
        if st is not None:
            if st["step"] is None and st["seen"] >= AUTO_AFTER and not st["failed"]:
                err = None
                try:
                    self._auto_make(st)
                except Exception as e:              # the step cannot be recorded (a stray torch operation, ...): stay eager
                    err = f"{type(e).__name__}: {e}"
                # every rank replays or every rank stays eager: a replay and an eager step issue the same collectives, but a
                # rank that failed half-way must not be left alone with the others' next exchange
                from . import dist as sdist
                err = sdist.agree_on_failure(err, _active_dist(), self.device)
                if err is not None:
                    st.update(step=None, full=None, aux=None, attrs=None, failed=err)
                    import warnings
                    warnings.warn(f"CSModel.update(): recording the step failed, staying eager ({err})")
            if st["step"] is not None:
                return self._auto_replay(st)
            st["seen"] += 1
        self.step_mode = "eager"
        self._materialize_input()
        return self._update_eager()

    @_own_arena
    def _update_eager(self):
        with ops.conv_precision(self._conv_mode()):
            return self._update()

    def _auto_state(self):
        """The auto-record state for the current configuration, or None when this call must run eagerly."""
        if (not AUTO_RECORD[0] or not getattr(self, "auto_record", True) or getattr(self, "_auto_busy", False) or not self.training
                or _lib.REC is not None or ops.TIMER is not None or self.device.type != "cuda" or not hasattr(self, "img_full")
                or self.cfg.reg not in ("None", "Rec") or torch.cuda.is_current_stream_capturing()
                or getattr(self, "_split_capture", False)):
            return None
        dist = _active_dist()
        pr = self.net_mask.pruned
        key = (tuple(self.img_full.shape), self.img_full.dtype, tuple(self.img_aux.shape), str(self.img_full.device), self.cfg.reg,
               float(self.cfg.weight_sim), float(self.cfg.weight_smooth), self._conv_mode(), self.bwd_dtype,
               dist.get_world_size() if dist is not None else 1, self.net_T.training, self.net_R.training,
               tuple(p.requires_grad for o in (self.optim_R, self.optim_T) for p in o._params()),
               tuple(o.bucket().flat_p.data_ptr() for o in (self.optim_R, self.optim_T)), pr.data_ptr(), pr._version,
               ops.F16_FWD[0], ops.F16_BWD[0], ops.USE_BF16X3[0], ops.WGRAD_OVERLAP[0], ops.WGRAD_DEFER[0], SENS_OVERLAP[0],
               bool(getattr(self, "time_exchange", False)), os.environ.get("SAN_GRAD_EXCHANGE", "allreduce"),
               # baked into the recording as host-side arguments: the low-frequency count of the sensitivity estimate, AdamW's
               # betas / eps (lr, weight decay and the step count live in device memory: sync_hyper)
               float(self.cfg.sparsity), int(self.cfg.shape),
               tuple((tuple(g["betas"]), float(g["eps"])) for o in (self.optim_R, self.optim_T) for g in o.param_groups))
        st = getattr(self, "_auto", None)
        if st is None or st["key"] != key:
            # a few recordings are kept by key (AUTO_KEEP): the short last batch of an epoch, or a validation pass with other
            # loss weights, does not throw the full batch's recording away
            cache = self.__dict__.setdefault("_auto_cache", {})
            if st is not None and st["step"] is not None:
                cache[st["key"]] = st
                while len(cache) > AUTO_KEEP:
                    cache.pop(next(iter(cache)))
            st = cache.pop(key, None) or {"key": key, "seen": 0, "step": None, "full": None, "aux": None, "attrs": None, "failed": None}
            self._auto = st
        return st

    def _auto_make(self, st) -> None:
        full, aux = self.img_full.clone(), self.img_aux.clone()
        st["step"] = self.record_update(full, aux, warmup=1, restore=True)
        st["full"], st["aux"] = full, aux
        # the recording's img_* / loss_* tensors: every replay refreshes them in place
        st["attrs"] = {k: v for k, v in self.__dict__.items() if k.startswith(("img_", "loss_", "metric_")) and k != "loss_all"}

    def _auto_replay(self, st) -> None:
        self._input_pending = False                     # (the replay's own prologue produces every img_* attribute)
        if self.img_full is not st["full"]:
            _lib.rec(st["full"].copy_, self.img_full)
        if self.img_aux is not st["aux"]:
            _lib.rec(st["aux"].copy_, self.img_aux)
        for o in (self.optim_R, self.optim_T):
            o.sync_hyper()                              # a learning-rate change since the last step reaches the device copy
        st["step"].replay()
        for name in [k for k in self.__dict__ if k.startswith(("loss_", "img_", "metric_"))]:
            delattr(self, name)
        self.__dict__.update(st["attrs"])
        self.step_mode = "replay of a recorded step (auto-recorded by update())"

    def _conv_mode(self) -> str:
        return self.conv_dtype or ("bf16" if self.use_amp else "bf16x3")

    def _update(self, part: str = "all"):
        """One optimisation step.  Regimes 'None' (train R, T frozen) and 'Rec' (train T and R through
        the warp), model.py:193-216; the GAN regimes are out of scope.  fp32 throughout: the
        GradScaler of the reference's AMP path is a no-op here.
        part: "all", or "front" (forward + backward [+ exchange launched inside a capture]) / "back" (exchange joined,
        optimiser steps) -- the two halves capture_update() records separately when the exchange cannot be captured."""
        assert self.training is True
        reg = self.cfg.reg
        if reg not in ("None", "Rec"):
            raise NotImplementedError(f"regime {reg!r}: the GAN branch (Mixed / GAN-Only) is out of scope")
        train_T = reg == "Rec"
        opts = [self.optim_R] + ([self.optim_T] if train_T else [])
        dist = _active_dist()
        if part in ("all", "front"):
            if dist is not None and not getattr(self, "_replicas_synced", False):
                # first data-parallel step: every rank starts from rank 0's state, and the inputs are rebuilt with rank 0's
                # column mask (set_input ran before the broadcast with this rank's own draw)
                self.sync_replicas()
                self.set_input(self.img_full, self.img_aux)
            self.loss_all = 0
            if train_T:
                self.forwardT()
            else:
                with torch.no_grad():
                    self.forwardT()
                self.loss_all = 0
            self.forwardR()
            for o in opts:
                o.zero_grad()                       # one memset of the flat gradient buffer per network
            from . import dist as sdist
            self._exchange = None
            if dist is not None and not getattr(self, "_split_capture", False):
                self._exchange = sdist.GradExchange(dist, timed=getattr(self, "time_exchange", False))
            with ops.conv_precision(self.bwd_dtype or self._conv_mode()):
                self.backward(train_T)              # weight gradients on a side stream, joined before the exchange / step
            if self._exchange is not None and train_T:
                self._exchange.launch(self.optim_T.bucket())
            del self.loss_all
        if part in ("all", "back"):
            exch = getattr(self, "_exchange", None)
            if exch is not None:
                exch.wait()
                self.exchange_slices = list(exch.launched)      # what went out, in order (None = a whole buffer): tests, bench
                if exch.events:
                    self._exchange_events = getattr(self, "_exchange_events", []) + exch.events
                if exch.wait_events:
                    self._exchange_wait_events = getattr(self, "_exchange_wait_events", []) + exch.wait_events
                self._exchange = None
            scale = 1.0 / dist.get_world_size() if dist is not None else 1.0   # the 1/world factor rides in the optimiser kernel
            for o in opts:
                o.step(grad_scale=scale)

    def _exchange_eager(self):
        """The gradient exchange between the two graphs of a split capture (gloo staging / a collective that refused to be
        captured): both buckets, on the main stream."""
        dist = _active_dist()
        if dist is None:
            return
        for o in [self.optim_R] + ([self.optim_T] if self.cfg.reg == "Rec" else []):
            o.bucket().allreduce_sum(dist)

    def exchange_ms(self, reset: bool = True) -> float:
        """Duration of the gradient all-reduces recorded since the last call (``time_exchange = True``; synchronise first)."""
        ev = getattr(self, "_exchange_events", [])
        ms = float(sum(e0.elapsed_time(e1) for e0, e1 in ev))
        if reset:
            self._exchange_events = []
        return ms

    def exchange_exposed_ms(self, reset: bool = True) -> float:
        """How long the main stream waited for the gradient exchange at its join since the last call (``time_exchange = True``;
        synchronise first): the part of ``exchange_ms`` the backward pass did NOT hide."""
        ev = getattr(self, "_exchange_wait_events", [])
        ms = float(sum(e0.elapsed_time(e1) for e0, e1 in ev))
        if reset:
            self._exchange_wait_events = []
        return ms

    @_no_auto
    def capture_update(self, img_full, img_aux=None, warmup: int = 3, restore: bool = True):
        """Capture ``set_input(img_full, img_aux); update()`` into a hipGraph and return an object whose ``replay()`` runs
        one optimisation step on whatever the two input tensors hold at that time (refill them in place between
        replays).  ~2,500 launches per step then cost one graph launch on the host -- the form to use when eight ranks
        share one host.  The step count and (lr, weight decay, gradient scale) of both optimisers move to device
        memory (FusedAdamW.device_step / sync_hyper: change ``param_groups[0]['lr']``, call ``optim.sync_hyper()``, and
        the next replay uses it); the weight gradients' side stream is captured as a fork / join.

        Under a process group the step is recorded as two graphs -- forward + backward, optimiser -- with the gradient
        exchange issued eagerly between them; ``replay()`` hides the difference (``.mode`` says which form it is).
        ``SAN_CAPTURE_COLLECTIVES=1`` tries to capture the RCCL all-reduces INSIDE one graph (see the note at ``split``).

        ``restore`` (default): parameters, AdamW moments, step counts and BatchNorm buffers are put back to their
        values from before the ``warmup`` real steps the capture needs (arena, packed weights, twiddles), so capturing
        does not train on duplicated data."""
        assert self.training is True
        dist = _active_dist()
        if dist is not None and not getattr(self, "_replicas_synced", False):
            self.sync_replicas()
        for o in (self.optim_R, self.optim_T):
            o.device_step = True
            o.bucket()                          # parameters move into the flat buffers now: their addresses are final
        snap = self._snapshot_state() if restore else None
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):          # warm-up off the default stream: arena, packs, twiddles, flat buffers exist
            for _ in range(max(1, warmup)):
                self.set_input(img_full, img_aux)
                self.update()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if snap is not None:
            self._restore_state(snap)
            torch.cuda.synchronize()
        pack_keep = []
        for reg in (ops.PACKS, ops.PACKS16):   # job tables are uploaded now, not inside the capture
            reg.ensure_table(self.device)
            pack_keep.append((reg.table, [j["packed"] for j in reg.order]))    # (the captured packing launch writes all of them)
        from . import dist as sdist

        def _capture(fn):
            graph = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may touch the runtime while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                fn()
            return graph

        def _whole():
            self.set_input(img_full, img_aux)
            self.update()

        # Under a process group the exchange is issued eagerly between two graphs by default.  Capturing the RCCL all-reduces (per
        # cascade, on the forked communication stream) inside ONE graph is opt-in (SAN_CAPTURE_COLLECTIVES=1): the first time this
        # path ran on RCCL (round 4, one rank, ROCm 7.0 / RCCL 2.26) hipStreamEndCapture crashed the process -- not an error
        # a caller could catch.  A lone captured all-reduce on the capturing stream does work (scratch/rccl_single.py).
        split = dist is not None and (sdist.backend() != "nccl" or os.environ.get("SAN_CAPTURE_COLLECTIVES", "0") != "1")
        if not split:
            try:
                step = CapturedStep([_capture(_whole)], None, "single-graph" + (" (RCCL all-reduce captured)" if dist is not None else ""))
                step.keep = pack_keep
                return step
            except RuntimeError:
                if dist is None:
                    raise
                split = True                    # the collective refused capture: record the two halves around it
                torch.cuda.synchronize()
        self._split_capture = True
        try:
            def _front():
                self.set_input(img_full, img_aux)
                with ops.use_arena(ops.owner_arena(self)), ops.conv_precision(self._conv_mode()):
                    self._update("front")

            def _back():
                with ops.use_arena(ops.owner_arena(self)), ops.conv_precision(self._conv_mode()):
                    self._update("back")

            g1, g2 = _capture(_front), _capture(_back)
        finally:
            self._split_capture = False
        step = CapturedStep([g1, g2], self._exchange_eager, "two graphs around an eager exchange")
        step.keep = pack_keep
        return step

    @_no_auto
    def record_update(self, img_full, img_aux=None, warmup: int = 2, restore: bool = True, timer=None):
        """Record ``set_input(img_full, img_aux); update()`` once and return an object whose ``replay()`` re-issues exactly
        that step's ~2,000 C-ABI calls, stream / event operations and the handful of torch operations as a flat loop over
        ``(callable, args)`` pairs (``_lib.rec``) -- the host then spends ~a third of what the eager step costs it, with
        ordinary stream semantics (ROCm's hipGraph launch was measured to cost the host as much as eager launching).
        Contract = a graph's: the two input tensors are static (refill them in place between replays), every tensor the
        step created stays alive inside the returned object and is overwritten by each replay (``img_*`` / ``loss_sim`` /
        ``loss_smooth`` keep pointing at them; ``loss_all`` is not recomputed), optimiser step counts and hyper-parameters
        live in device memory (``optim.sync_hyper()`` after changing the learning rate).  Works under a process group (the
        collectives are part of the recording).  ``restore``: the warm-up steps and the recorded step itself are undone.
        ``timer``: an ops.KernelTimer active during the recorded step only -- its event brackets become part of the recording
        (``timer.totals(replays=K)`` after K replays)."""
        assert self.training is True and ops.TIMER is None
        dist = _active_dist()
        if dist is not None and not getattr(self, "_replicas_synced", False):
            self.sync_replicas()
        for o in (self.optim_R, self.optim_T):
            o.device_step = True
            o.bucket()
        snap = self._snapshot_state() if restore else None

        def run():
            self.set_input(img_full, img_aux)
            self.update()

        try:
            for _ in range(max(1, warmup)):     # arenas, packed weights, twiddles, flat buffers, constant tables exist afterwards
                run()
            torch.cuda.synchronize()
            self._exchange_events = []           # (only the recorded step's exchange events are kept)
            self._exchange_wait_events = []
            step = self._record(run, "record_update", timer)
        finally:
            # ALSO when the warm-up or the recording raised (a stray torch operation, out of memory): the caller falls back to
            # the eager step, which must start from the state before the warm-up -- not one to three optimiser steps later
            if snap is not None:
                torch.cuda.synchronize()
                self._restore_state(snap)
                torch.cuda.synchronize()
        return step

    @_no_auto
    def record_forward(self, img_full, img_aux=None, warmup: int = 1):
        """The inference pass (``set_input; forwardT; forwardR`` under ``no_grad``, as ``test()`` runs it minus the host-side
        metrics) as a recorded step: ``replay()`` refreshes ``img_rec`` / ``img_warped`` / ``loss_sim`` ... in place for
        whatever the two static input tensors hold."""
        def run():
            with torch.no_grad(), ops.conv_precision(self._conv_mode()):
                self.set_input(img_full, img_aux)
                self.loss_all = 0
                self.forwardT()
                self.loss_all = 0
                self.forwardR()

        for o in (self.optim_R, self.optim_T):
            o.bucket()                          # parameters move into the flat buffers now: the recording holds their final addresses
        for _ in range(max(1, warmup)):
            run()
        torch.cuda.synchronize()
        ops.bump_weight_epoch()                 # the recorded pass then contains the batched weight-packing launches: a replay
        return self._record(run, "record_forward", None)     # after an optimiser step / load() re-packs (RecordedStep.replay)

    def _record(self, run, what: str, timer):
        """Execute ``run()`` under the recorder (``_lib.REC``) and a dispatch mode that keeps every tensor it creates alive
        and refuses stray torch operations; returns the RecordedStep."""
        from torch.utils._python_dispatch import TorchDispatchMode
        keep, stray = [], []
        for reg in (ops.PACKS, ops.PACKS16):   # the pack job tables are uploaded now (a host-to-device copy), not inside the step
            reg.ensure_table(self.device)
            keep.append(reg.table)              # the recorded packing launch reads THIS table: it must outlive a later re-registration
            # ... and it WRITES every packed image the table names, other live models' included: those buffers stay allocated as
            # long as this recording lives.  (Before round 4's end only the table was kept: once another model was freed and its
            # jobs pruned, a replay wrote its packed weights into memory the allocator had handed to somebody else -- the likely
            # source of the one-in-twenty bit-identity failures in long test processes.)
            keep.append([j["packed"] for j in reg.order])

        class _Watch(TorchDispatchMode):
            """Keeps every tensor the step creates alive (their addresses are in the recording) and notes torch operations
            with side effects on device memory that neither went through _lib.rec nor were declared untracked."""
            QUIET = ("aten.empty", "aten.view", "aten._unsafe_view", "aten.reshape", "aten.permute", "aten.select", "aten.slice",
                     "aten.detach", "aten.alias", "aten.expand", "aten.as_strided", "aten.t.", "aten.transpose", "aten.unsqueeze",
                     "aten.squeeze", "aten.view_as_real", "aten.view_as_complex", "aten.record_stream", "aten.new_empty",
                     "aten.is_pinned", "profiler.", "aten.sym_", "aten.size", "aten.stride", "aten.storage_offset",
                     "aten.is_contiguous", "aten.numel", "aten.dim", "aten.unbind", "aten.split", "aten.chunk", "aten.lift_fresh")

            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                out = func(*args, **(kwargs or {}))
                keep.append(out)
                name = str(func)
                if not _lib.IN_REC[0] and not _lib.UNTRACKED[0] and not name.startswith(self.QUIET):
                    import os
                    import traceback
                    site = next((f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in reversed(traceback.extract_stack(limit=14))
                                 if "spatialalignmentnetwork_amd" in fr.filename and not fr.name.startswith("__torch_dispatch__")), "?")
                    stray.append(f"{name} @ {site}")
                return out

        _lib.REC, _lib.KEEP, _lib.LIGHT = [], keep, set()
        ops.TIMER = timer
        try:
            with _Watch():
                run()
        finally:
            calls, light, _lib.REC, _lib.KEEP, _lib.LIGHT = _lib.REC, _lib.LIGHT, None, None, None
            ops.TIMER = None
        torch.cuda.synchronize()
        if stray:
            raise RuntimeError(f"{what}: torch operations outside _lib.rec / _lib.untracked in the step (a host "
                               "synchronisation such as .item() counts): " + ", ".join(sorted(set(stray))))
        step = RecordedStep(calls, keep, training=what == "record_update")
        step.light = light                      # ids of the events that only order this GPU's streams (raw, fence-free in the tapes)
        return step

    def _state_tensors(self):
        ts = []
        for o in (self.optim_R, self.optim_T):
            b = o.bucket()
            ts += [b.flat_p, b.exp_avg, b.exp_avg_sq]
            if b.step_dev is not None:
                ts.append(b.step_dev)
        for mod in (self.net_T, self.net_R):
            ts += list(mod.buffers())
        return ts

    def _snapshot_state(self):
        return ([t.detach().clone() for t in self._state_tensors()], [o.bucket().steps for o in (self.optim_R, self.optim_T)],
                [o.bucket().step_dev is not None for o in (self.optim_R, self.optim_T)])

    def _restore_state(self, snap):
        saved, steps, had_dev = snap
        it = iter(saved)
        for o, st, hd in zip((self.optim_R, self.optim_T), steps, had_dev):
            b = o.bucket()
            for t in (b.flat_p, b.exp_avg, b.exp_avg_sq):
                t.copy_(next(it))
            if hd:
                b.step_dev.copy_(next(it))
            elif b.step_dev is not None:
                b.step_dev.fill_(st)            # created by the warm-up steps: back to the host count
            b.steps = st
        for mod in (self.net_T, self.net_R):
            for buf in mod.buffers():
                buf.copy_(next(it))
        ops.bump_weight_epoch()                 # packed weight images are stale

    def sync_replicas(self, dist=None) -> None:
        """Broadcast rank 0's parameters, AdamW moments, step counts, BatchNorm buffers and column mask to every rank
        (what DDP does at construction).  Without it replicas built from per-process RNG streams would average
        gradients of DIFFERENT models.  Call it BEFORE the first set_input (the sampling mask it broadcasts shapes the
        inputs); update() calls it on the first data-parallel step if nobody did, and then re-runs set_input.  Call it
        again after load()."""
        dist = dist or _active_dist()
        if dist is None:
            return
        from . import dist as sdist
        for o in (self.optim_R, self.optim_T):
            b = o.bucket()
            for t in (b.flat_p, b.exp_avg, b.exp_avg_sq):
                sdist.broadcast0(t, dist)
            have = b.steps if b.step_dev is None else int(b.step_dev.item())
            steps = torch.tensor([have], dtype=torch.int64, device=b.flat_p.device)
            sdist.broadcast0(steps, dist)
            b.steps = int(steps.item())
            if b.step_dev is not None:
                b.step_dev.fill_(b.steps)
        for mod in (self.net_T, self.net_R, self.net_mask):
            for buf in mod.buffers():
                sdist.broadcast0(buf, dist)
        sdist.broadcast0(self.net_mask.weight.data, dist)
        ops.bump_weight_epoch()                 # packed weight images are stale
        _KEEP_CACHE.clear()
        self._replicas_synced = True

    def _cascade_ranges(self, bucket):
        """{cascade index | "sens": (lo, hi) of net_R's flat gradient buffer}; together they cover the whole buffer."""
        ranges = {j: bucket.range_of(list(c.parameters())) for j, c in enumerate(self.net_R.cascades)}
        ranges["sens"] = bucket.range_of(list(self.net_R.sens_net.parameters()))
        covered = sorted(r for r in ranges.values() if r is not None)
        pos = 0
        for lo, hi in covered:
            assert lo == pos, "net_R's parameters are not partitioned by (sens_net, cascades)"
            pos = hi
        assert pos == bucket.total, (pos, bucket.total)
        return ranges

    def _grad_buckets(self):
        """Flat per-network buffers (p.data / p.grad are views, see dist.ParamBucket)."""
        return {"R": self.optim_R.bucket(), "T": self.optim_T.bucket()}

    @_own_arena
    def test(self):
        """model.py:265-286 without the GAN branch; returns -PSNR."""
        assert self.training is False
        with torch.no_grad(), ops.conv_precision(self._conv_mode()):
            self.loss_all = 0
            self.forwardT()
            self.loss_all = 0
            self.forwardR()
            # model.py:275-279, on the GPU, one host synchronisation for all five scalars
            m = metrics.test_metrics(self.img_full_rss, self.img_rec, self.img_warped_rss)
            self.metric_MI = m["MI"]
            self.metric_PSNR = m["PSNR"]
            self.metric_SSIM = m["SSIM"]
            self.metric_MAE = m["MAE"]
            self.metric_MSE = m["MSE"]
        return -self.metric_PSNR

    def get_vis(self, content=None):
        """model.py:292-321."""
        assert content in [None, "scalars", "histograms", "images"]
        self._materialize_input()                       # (a set_input whose derived images a replay has not produced yet)
        vis = {}
        if content in (None, "scalars"):
            vis["scalars"] = {}
            for k, v in self.__dict__.items():
                if k.startswith("loss_") and v is not None:
                    # incl. loss_all, which test() leaves behind (= loss_sim * weight_sim, model.py:270-272,296-300);
                    # update() deletes it (model.py:262)
                    vis["scalars"][k] = v.detach().item()
                elif k.startswith("metric_") and v is not None:
                    vis["scalars"][k] = v
        if content in (None, "images"):
            vis["images"] = {k: v.detach() for k, v in self.__dict__.items()
                             if k.startswith("img_") and v is not None and not torch.is_complex(v)
                             and v.shape[1] in (1, 3)}
        if content in (None, "histograms"):
            vis["histograms"] = {"weights": {"values": self.net_mask.weight.detach()}}
        return vis


class RecordedStep:
    """What CSModel.record_update / record_forward return: ``replay()`` re-issues the recorded step.

    Every C-ABI call of a replay has its return code checked (a failed launch raises, as it does in the eager step).
    ``training`` steps change the weights, so a replay ends with ``ops.bump_weight_epoch()``: the next eager forward re-packs
    its weight images (the optimiser's own bump is host code and not part of the recording).  Forward-only recordings
    contain the batched weight-packing launches and run them only when the weights changed since their last replay."""

    # Optional run-ahead throttle (SAN_REPLAY_CHUNK > 0; default off): every CHUNK calls the replay records a blocking event
    # (hipEventBlockingSync) and, before going on, sleeps on the one recorded LAG chunks earlier, so that at most LAG * CHUNK calls
    # are ever queued.  Built to test whether the ~1.8 busy host cores per rank of a replayed step (host_cpu_ms ~ 84 per 46 ms
    # step) are the runtime spinning on a full launch queue: they are NOT -- 128..1024 x 2..3 calls in flight leave
    # host_cpu_ms at 80-88 and the step time unchanged (scratch/attempts/r4_host_depth.txt); the second busy thread is the
    # runtime's own.  Kept as a hook for boxes with fewer cores than 2 per rank.
    CHUNK = int(os.environ.get("SAN_REPLAY_CHUNK", "0"))
    LAG = int(os.environ.get("SAN_REPLAY_LAG", "3"))

    def __init__(self, calls, keep, training: bool = True):
        self.calls, self.keep, self.training = calls, keep, training
        self.mode = f"recorded step: {len(calls)} calls"
        self._packed_epoch = None
        self._ring, self._k = [], 0
        self._segs, self._own_events = None, []
        self.light = set()
        self._raw = None

    def _throttle(self) -> None:
        if not self._ring:
            self._ring = [torch.cuda.Event(blocking=True) for _ in range(self.LAG + 1)]
        ring, k = self._ring, self._k
        ring[k % len(ring)].record()
        if k >= self.LAG:
            ring[(k - self.LAG) % len(ring)].synchronize()      # the host sleeps until the GPU is within LAG chunks
        self._k = k + 1

    # SAN_NATIVE_REPLAY=0: the Python loop below (one ctypes call per recorded entry), kept for comparison and for the throttle
    NATIVE = os.environ.get("SAN_NATIVE_REPLAY", "1") != "0"

    def _compile(self):
        """The recording as segments: runs of C-ABI calls and stream / event operations become tapes for san_replay_run (one
        foreign call each), whatever else was recorded (tensor copies, the collectives, a few torch expressions) stays a Python
        callable between them.  Order and arguments are the recorded ones; ``wait_stream`` gets an event of its own, as torch
        does per call."""
        segs, words, names = [], [], {}
        own = []                                # events made for wait_stream: alive as long as the step
        # raw events without the system-scope fence for the step's own stream hand-offs (SAN_LIGHT_EVENTS=0: torch's events everywhere).
        # NOT for the gradient exchange: peers read and write this GPU's memory around the collective, so every
        # hand-off that names a communication stream keeps a system-scope event.
        use_raw = os.environ.get("SAN_LIGHT_EVENTS", "1") != "0"
        raw, raw_of = (_lib.RawEvents() if use_raw else None), {}
        from . import dist as _sdist
        comm_streams = {st.cuda_stream for st in _sdist.GradExchange._streams.values()}

        def raw_event(stream, light=True):
            """A raw handle on ``stream``'s device, or None when the library cannot make one (then torch's events serve)."""
            nonlocal use_raw
            if not use_raw:
                return None
            try:
                return raw.new(stream.device.index if stream.device.index is not None else torch.cuda.current_device(), light)
            except Exception as e:          # (a failing helper must not take the replay down: torch's events always work)
                print(f"[san] raw events unavailable ({type(e).__name__}: {e}); using torch events", flush=True)
                use_raw = False
                return None

        # (decided per EVENT, before any handle is made: an event that a communication stream records or waits for anywhere in
        # the step is torch's own on both sides)
        heavy = set()
        for fn, args, kind in self.calls:
            owner, name = getattr(fn, "__self__", None), getattr(fn, "__name__", "")
            if kind == 0 and len(args) == 1:
                if isinstance(owner, torch.cuda.Event) and name == "record" and isinstance(args[0], torch.cuda.Stream) \
                        and args[0].cuda_stream in comm_streams:
                    heavy.add(id(owner))
                elif isinstance(owner, torch.cuda.Stream) and name == "wait_event" and owner.cuda_stream in comm_streams:
                    heavy.add(id(args[0]))

        def handle(ev, stream):
            if use_raw and id(ev) in self.light and id(ev) not in heavy:
                h = raw_of.get(id(ev))
                if h is None:
                    h = raw_event(stream)
                    if h is None:
                        return ev.cuda_event
                    raw_of[id(ev)] = h
                return h
            return ev.cuda_event

        def flush():
            if words:
                segs.append(_lib.Tape(list(words), dict(names)))
                words.clear()
                names.clear()

        def ev_record(ev, stream):
            names[len(words)] = "hipEventRecord"
            words.extend((_lib.TAPE_EVENT_RECORD | 2 << 24, ev if isinstance(ev, int) else handle(ev, stream), stream.cuda_stream))

        def st_wait(stream, ev):
            names[len(words)] = "hipStreamWaitEvent"
            words.extend((_lib.TAPE_STREAM_WAIT | 2 << 24, stream.cuda_stream, ev if isinstance(ev, int) else handle(ev, stream)))

        for fn, args, kind in self.calls:
            enc = None
            if kind:
                enc = _lib.tape_call_words(fn, args, _lib.TAPE_PACK if kind == 2 else 0)
            elif isinstance(fn, ctypes._CFuncPtr):
                enc = _lib.tape_call_words(fn, args, _lib.TAPE_IGNORE_RC)       # (san_wgrad_defer returns the previous mode)
            else:
                owner, name = getattr(fn, "__self__", None), getattr(fn, "__name__", "")
                if isinstance(owner, torch.cuda.Event) and name == "record" and len(args) == 1 and isinstance(args[0], torch.cuda.Stream) \
                        and owner.cuda_event:
                    ev_record(owner, args[0])
                    continue
                if isinstance(owner, torch.cuda.Stream) and name == "wait_event" and len(args) == 1 and isinstance(args[0], torch.cuda.Event) \
                        and args[0].cuda_event:
                    st_wait(owner, args[0])
                    continue
                if isinstance(owner, torch.cuda.Stream) and name == "wait_stream" and len(args) == 1 and isinstance(args[0], torch.cuda.Stream):
                    # (an event of the recorded stream's device; with a communication stream on either side: system scope)
                    peer = owner.cuda_stream in comm_streams or args[0].cuda_stream in comm_streams
                    ev = raw_event(args[0], light=not peer)
                    if ev is None:
                        ev = torch.cuda.Event()
                        ev.record(args[0])      # (creates the handle; harmless: nothing waits for this one)
                        own.append(ev)
                    ev_record(ev, args[0])
                    st_wait(owner, ev)
                    continue
            if enc is None:
                if kind:
                    raise RuntimeError(f"recorded step: {getattr(fn, '__name__', fn)} cannot be put on a replay tape")
                flush()
                segs.append((fn, args))
            else:
                names[len(words)] = getattr(fn, "__name__", "?")
                words.extend(enc)
        flush()
        self._own_events = own
        self._raw = raw                         # (the tapes hold these handles: alive as long as the step)
        return segs

    def invalidate(self) -> None:
        """Forget the tapes: the next replay rebuilds them from ``calls`` (tests edit a recorded call)."""
        self._segs, self._own_events, self._raw = None, [], None

    def replay(self) -> None:
        skip_packs = (not self.training) and self._packed_epoch == ops.WEIGHT_EPOCH[0]
        if self.NATIVE and not self.CHUNK:
            if self._segs is None:
                self._segs = self._compile()
            for seg in self._segs:
                if type(seg) is tuple:
                    seg[0](*seg[1])
                else:
                    seg.run(skip_packs)
        else:
            self._replay_python(skip_packs)
        if self.training:
            ops.bump_weight_epoch()
        else:
            # (the registries' own epoch is left alone: the recording's job table covers the jobs that existed when it was made,
            # an eager step may have registered more since -- data-gradient images -- and re-packs all of them itself)
            self._packed_epoch = ops.WEIGHT_EPOCH[0]

    def _replay_python(self, skip_packs: bool) -> None:
        chunk, left = self.CHUNK, self.CHUNK
        for fn, args, kind in self.calls:
            if kind:
                if kind == 2 and skip_packs:
                    continue
                rc = fn(*args)
                if rc:
                    raise RuntimeError(f"recorded step: {getattr(fn, '__name__', fn)} failed ({'argument error' if rc < 0 else 'hipError_t'} "
                                       f"{rc}): {_lib.lib().last_error()}")
                if chunk:
                    left -= 1
                    if left == 0:
                        left = chunk
                        self._throttle()
            else:
                fn(*args)


class CapturedStep:
    """What CSModel.capture_update returns: ``replay()`` runs one optimisation step."""

    def __init__(self, graphs, between, mode: str):
        self.graphs, self.between, self.mode = graphs, between, mode

    def replay(self) -> None:
        self.graphs[0].replay()
        for g in self.graphs[1:]:
            if self.between is not None:
                self.between()
            g.replay()
        ops.bump_weight_epoch()                 # the captured optimiser step changed the weights (its host-side bump is not in the graph)


def _active_dist():
    import torch.distributed as dist
    from . import dist as sdist
    return dist if (dist.is_available() and dist.is_initialized()
                    and (dist.get_world_size() > 1 or sdist.single_rank_exchange())) else None


_KEEP_CACHE = {}


def _keep_mask(pruned: torch.Tensor) -> torch.Tensor:
    """float [W] 1 = sampled; cached per buffer (the mask is fixed for a model's lifetime)."""
    key = (pruned.data_ptr(), pruned._version, str(pruned.device))
    hit = _KEEP_CACHE.get(key)
    if hit is None or hit[0] is not pruned:
        hit = (pruned, (~pruned).to(torch.float32).contiguous())
        _KEEP_CACHE[key] = hit
    return hit[1]
