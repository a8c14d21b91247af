"""Tensor-level wrappers over the C ABI (include/san_hip.h).

PyTorch is plumbing here: it owns device memory and the stream.  Every function
below hands raw device pointers + the current HIP stream to libsan_hip.so; no
ATen compute kernel is launched on the hot path.
"""
from __future__ import annotations

import ctypes
import os
import weakref
import dataclasses
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import lib

NORM_INSTANCE, NORM_GROUP, NORM_BATCH, NORM_GROUP_BWD = 0, 1, 2, 3


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (bench.py
    uses it for the roofline figures).  bracket(name, work, unit, fn) times one
    C-ABI call; totals() needs a prior device synchronisation.

    An event pair costs a few microseconds and keeps neighbouring kernels from
    overlapping (and, inside wgrad_overlap, joins the two streams), so only every
    ``stride``-th launch of a family is bracketed (a prime stride walks through
    all layers of the periodic cascade structure);
    every launch is still counted, and totals() scales the sampled time up."""

    def __init__(self, stride: int = 29, strides: Optional[dict] = None, alone: bool = True):
        self.stride = max(1, int(stride))
        # alone: inside wgrad_overlap a bracketed launch runs ALONE (the other stream is joined before and held until after it: the
        # kernel's own time, at the price of draining the weight-gradient queue at every bracket).  False (bench.py since round 6): the
        # brackets only mark the launch on its own stream -- the time AS SHIPPED, beside whatever the other stream runs, which is what
        # rocprofv3 reports for the same command
        self.alone = bool(alone)
        self.strides = dict(strides or {})      # per-family override (bench.py brackets EVERY cascade-boundary launch)
        self.recs = []
        self.seen = {}
        self.last = {}                          # family -> (fn, work) of its most recent launch (batch())

    def bracket(self, name, work, unit, fn, abytes=0.0, xwork=0.0):
        d = self.seen.setdefault(name, {"launches": 0, "work": 0.0, "unit": unit, "abytes": 0.0, "xwork": 0.0})
        d["launches"] += 1
        d["work"] += work
        d["abytes"] += abytes
        d["xwork"] += xwork
        self.last[name] = (fn, work)
        if (d["launches"] - 1) % self.strides.get(name, self.stride):
            fn()
            return
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        # inside wgrad_overlap two streams share the GPU: a bracketed launch is run alone (the other stream is joined
        # before and held until after it), so that the time is the kernel's own and not the contention's
        side, other = _WG["stream"], None
        cur = _launch_stream()
        # (through _lib.rec: inside a recorded step the brackets are part of the recording, every replay re-records the
        # same event pairs and totals() reads the last replay's)
        if side is not None and self.alone:
            other = _WG["main"] if cur == side else side
            _lib.rec(cur.wait_stream, other)
        _lib.rec(e0.record, cur)
        fn()
        _lib.rec(e1.record, cur)
        if other is not None:
            _lib.rec(other.wait_stream, cur)
        self.recs.append((name, work, e0, e1))

    def batch(self, name: str, count: int = 12, rounds: int = 5):
        """ONE event pair around ``count`` back-to-back re-issues of the family's most recent launch (same arguments, on the
        current stream, nothing else running): the pair's own few microseconds amortise over the batch.  Returns
        (average microseconds per launch -- the best of ``rounds`` batches --, work per launch) or None."""
        if name not in self.last:
            return None
        fn, work = self.last[name]
        best = None
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(count):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / count
            best = t if best is None else min(best, t)
        return best, work

    @staticmethod
    def event_pair_overhead_us(pairs: int = 64) -> float:
        """Median elapsed time of an event pair with NOTHING between the two records, on the current stream: the part of
        every bracket that is the measurement's own (reported next to the raw figures, never subtracted from them)."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(pairs)]
        torch.cuda.synchronize()
        for e0, e1 in evs:
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
        return float(t[len(t) // 2])

    def totals(self, replays: int = 1):
        """{family: launches, work (all launches), sampled_launches / sampled_ms / sampled_work (the bracketed ones),
        ms = sampled_ms scaled by work (the estimate for all launches)}.  replays: the timer saw ONE recorded step that was
        then replayed this many times (launches / work / ms are scaled to the whole timed region; the sampled figures are
        the last replay's event pairs)."""
        out = {k: dict(v, sampled_launches=0, sampled_ms=0.0, sampled_work=0.0) for k, v in self.seen.items()}
        for d in out.values():
            for key in ("launches", "work", "abytes", "xwork"):
                d[key] = d[key] * replays
        for name, work, e0, e1 in self.recs:
            d = out[name]
            d["sampled_launches"] += 1
            d["sampled_ms"] += e0.elapsed_time(e1)
            d["sampled_work"] += work
        for d in out.values():
            d["ms"] = d["sampled_ms"] * d["work"] / d["sampled_work"] if d["sampled_work"] > 0 else 0.0
        return out


TIMER: Optional[KernelTimer] = None


TIMED_KERNELS = ("conv3x3", "conv3x3_bf16x3", "wgrad3x3", "wgrad3x3_bf16x3", "fft_dc", "fft_dc_bwd", "act_bwd")     # event pairs serialise neighbouring kernels: time only what the roofline needs


def _timed(name, work, unit, fn, abytes=0.0, products=0):
    """abytes: the launch's ALGORITHMIC bytes (operands read once + results written once), reported next to the PMC
    traffic so that a waste ratio can be formed.  products: matrix-core products the kernel executes per algorithmic MAC
    (6 / 3 / 1 bf16 parts modes, 3 for the fp16 two-part forward form)."""
    if TIMER is None or name not in TIMED_KERNELS:
        fn()
    else:
        TIMER.bracket(name, work, unit, fn, abytes, work * products)


def _products(f16: bool = False) -> int:
    return 3 if (f16 and _CONV_NP[0] == 3) else {3: 6, 2: 3, 1: 1}[_CONV_NP[0]]


def _conv_abytes(n, h, w, cin, cout, ks):
    """fp32 input planes + fp32 output planes + fp32 weights, each once."""
    return 4.0 * (n * h * w * (cin + cout) + cout * cin * ks * ks)


# Raw handle of the stream the next launch goes to.  torch.cuda.current_stream() resolves the device through several Python
# layers (~3 us; ~2,500 launches per training step), the two C calls below take ~0.3 us.  _STREAM_OVERRIDE: the weight-gradient
# side stream while _on_side_stream runs a launch there (its handle is passed to the C ABI directly instead of switching
# torch's current stream around every weight gradient).
_STREAM_OVERRIDE = [None]


def _stream() -> int:
    ov = _STREAM_OVERRIDE[0]
    if ov is not None:
        return ov.cuda_stream
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _launch_stream() -> "torch.cuda.Stream":
    """The stream object launches currently go to (the side stream inside _on_side_stream, else torch's current stream)."""
    return _STREAM_OVERRIDE[0] if _STREAM_OVERRIDE[0] is not None else torch.cuda.current_stream()


def _p(t: Optional[torch.Tensor]):
    """Device pointer argument of a C-ABI call: the address as a plain int (ctypes converts it for the c_void_p parameter;
    ~13,000 of these per training step, a c_void_p object each was measurable), None -> NULL."""
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype=torch.float32, name: str = "tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


def _creal(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    """A complex64 tensor checked for the C ABI (interleaved (re, im) fp32 = torch.view_as_real: the same storage and the
    same data pointer, so the tensor itself is handed on; building the real view was ~190 dispatches per training step)."""
    if t.dtype != torch.complex64:
        raise RuntimeError(f"{name} must be complex64, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (the HIP path has no CPU fallback)")
    return t


# ---------------------------------------------------------------------------
# scratch arena: named, shape-keyed device buffers reused across calls (the
# library never allocates activations; the caching allocator does, once)
# ---------------------------------------------------------------------------
# data pointers of buffers an arena owns (they are never handed back to the caching allocator while the arena lives, so a side
# stream that reads them needs no record_stream: 650 dispatches per training step)
_ARENA_PTRS = set()


_POISON = os.environ.get("SAN_ARENA_POISON", "0") == "1"


# Debugging hook (SAN_ARENA_GUARD=1, or ops.ARENA_GUARD[0] = True before the buffers are made): every arena buffer sits between two
# 4 KiB bands of a byte pattern (SAN_ARENA_GUARD_BYTE, hex: FF makes the bands NaN, so that a kernel that READS past its input and
# uses the value shows up in the results); arena_guard_report() names the buffers whose bands were written.  A kernel that stores
# past the end (or in front) of its output does no visible harm while the step's streams run one after the other, and corrupts a
# neighbour's data once they overlap.
ARENA_GUARD = [os.environ.get("SAN_ARENA_GUARD", "0") == "1"]
_GUARD_BYTES, _GUARD_PATTERN = 4096, int(os.environ.get("SAN_ARENA_GUARD_BYTE", "A5"), 16)
_GUARDS = []


def _guarded(key, shape, device, dtype, zero: bool) -> torch.Tensor:
    numel = 1
    for d in shape:
        numel *= int(d)
    nbytes = numel * torch.empty((), dtype=dtype).element_size()
    raw = torch.full((nbytes + 2 * _GUARD_BYTES,), _GUARD_PATTERN, dtype=torch.uint8, device=device)
    t = raw[_GUARD_BYTES:_GUARD_BYTES + nbytes].view(dtype).view(tuple(int(d) for d in shape))
    if zero:
        t.zero_()
    _GUARDS.append((key, raw, nbytes))
    return t


def arena_guard_report():
    """[(buffer name, shape), 'front' | 'back', number of overwritten guard bytes, byte offset of the nearest one from the buffer's
    start (negative) / end] for every guarded arena buffer whose bands no longer hold the pattern."""
    out = []
    torch.cuda.synchronize()
    for key, raw, nbytes in _GUARDS:
        for side, band in (("front", raw[:_GUARD_BYTES]), ("back", raw[_GUARD_BYTES + nbytes:])):
            hit = (band != _GUARD_PATTERN).nonzero()
            if hit.numel():
                first = int(hit[0]) if side == "back" else int(hit[-1]) - _GUARD_BYTES
                out.append((key[:2], side, int(hit.numel()), first))
    return out


class Arena:
    def __init__(self):
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self._retired = []

    def get(self, name: str, shape, device, dtype=torch.float32, zero: bool = False, _no_wait: bool = False) -> torch.Tensor:
        # (shapes arrive as tuples of Python ints or torch.Size: both hash / compare equal to the int tuple)
        key = (name, shape if type(shape) is tuple else tuple(shape), _DEV_KEYS.get(device) or _dev_key(device), dtype)
        t = self._bufs.get(key)
        if t is None:
            if ARENA_GUARD[0]:
                t = _guarded(key, shape, device, dtype, zero)
            else:
                t = torch.zeros(shape, device=device, dtype=dtype) if zero else torch.empty(shape, device=device, dtype=dtype)
            if _POISON and _lib.REC is not None:
                print("arena buffer first created inside a recording:", key[:2], flush=True)
            elif _POISON and not zero and t.is_floating_point():
                t.fill_(float("nan"))           # (debugging: a kernel that reads an arena buffer it never wrote shows up as NaN)
            self._bufs[key] = t
            _ARENA_PTRS.add(t.data_ptr())
        elif not _no_wait and (_WG["busy"] or _WG["pending_ptrs"]):
            _wait_if_busy(t)                    # a side-stream weight gradient may still be reading it (wgrad_overlap)
        return t

    def scratch(self, name: str, nbytes: int, device) -> torch.Tensor:
        """One grow-only byte buffer per name (launches on one stream are ordered, so successive users may share it)."""
        key = (name, "scratch", _dev_key(device))
        t = self._bufs.get(key)
        if t is None or t.numel() < nbytes:
            if t is not None:
                self._retired.append(t)         # a side-stream kernel may still be using it: never hand its memory back
            t = _guarded(key, (int(nbytes),), device, torch.uint8, False) if ARENA_GUARD[0] else torch.empty(int(nbytes), device=device, dtype=torch.uint8)
            self._bufs[key] = t
        return t

    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._bufs.values())

    def clear(self):
        for t in self._bufs.values():
            _ARENA_PTRS.discard(t.data_ptr())
        self._bufs.clear()

    def __del__(self):                          # an arena dies with its owner: its addresses may come back as ordinary tensors
        try:
            for t in self._bufs.values():
                _ARENA_PTRS.discard(t.data_ptr())
        except Exception:
            pass


_DEV_KEYS: Dict[object, str] = {}


def _dev_key(device) -> str:
    """'cuda' and 'cuda:0' name the same device: buffers and pools are keyed by the resolved form (memoised: this sits on
    the path of every arena request, thousands per training step)."""
    try:
        return _DEV_KEYS[device]
    except KeyError:
        pass
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        return str(torch.device("cuda", torch.cuda.current_device()))      # depends on the current device: not memoised
    _DEV_KEYS[device] = str(d)
    return _DEV_KEYS[device]


# Arenas have OWNERS.  Every CSModel (and every VarNet / SpatialTransformer used on its own) keeps its activation tapes in an
# arena of its own, entered for the duration of its forward / backward (``use_arena``): a second model in the process (a
# validation copy, an EMA) can run between A.forward and A.backward without touching A's tape.  Code that asks
# ``GLOBAL_ARENA`` gets whichever arena is current (the process-wide default outside any owner's scope).  Buffers of shapes
# that are no longer used stay allocated until Arena.clear() (the last partial batch of an epoch costs a second set of
# per-cascade buffers; 288 GB of HBM make that a non-issue for this workload).
_DEFAULT_ARENA = Arena()
_ARENA_STACK = [_DEFAULT_ARENA]


class _CurrentArena:
    """Forwards to the arena of the innermost ``use_arena`` scope."""

    def get(self, *a, **k):
        return _ARENA_STACK[-1].get(*a, **k)

    def scratch(self, *a, **k):
        return _ARENA_STACK[-1].scratch(*a, **k)

    def bytes(self) -> int:
        return _ARENA_STACK[-1].bytes()

    def clear(self):
        _ARENA_STACK[-1].clear()


GLOBAL_ARENA = _CurrentArena()


class use_arena:
    """``with use_arena(owner_arena):`` -- arena requests inside the block go to ``owner_arena``.  ``outer_only``: keep the
    current arena when some owner's scope is already active (a VarNet inside a CSModel uses the CSModel's arena)."""

    def __init__(self, arena: Arena, outer_only: bool = False):
        self.arena, self.outer_only = arena, outer_only

    def __enter__(self):
        top = _ARENA_STACK[-1]
        _ARENA_STACK.append(top if (self.outer_only and top is not _DEFAULT_ARENA) else self.arena)
        return _ARENA_STACK[-1]

    def __exit__(self, *exc):
        _ARENA_STACK.pop()
        return False


def owner_arena(owner) -> Arena:
    """The arena that belongs to ``owner`` (created on first use)."""
    a = owner.__dict__.get("_san_arena")
    if a is None:
        a = Arena()
        object.__setattr__(owner, "_san_arena", a)
    return a


@dataclass
class Act:
    """A lazily normalised activation: channels [coff, coff+c) of ``buf``
    ([N, ctot, H, W] raw values) to be read as lrelu(scale*x + shift, slope).
    ``scale``/``shift`` are [N, ctot] (same channel layout as ``buf``) or None."""
    buf: torch.Tensor
    coff: int
    c: int
    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    slope: float = 1.0
    amax: Optional[torch.Tensor] = None     # gradients: the tensor's amax record (int32 words, see AmaxPool)

    @property
    def n(self):
        return self.buf.shape[0]

    @property
    def ctot(self):
        return self.buf.shape[1]

    @property
    def h(self):
        return self.buf.shape[2]

    @property
    def w(self):
        return self.buf.shape[3]

    def view(self, coff: int, c: int) -> "Act":
        return Act(self.buf, self.coff + coff, c, self.scale, self.shift, self.slope, self.amax)


def full(buf: torch.Tensor, scale=None, shift=None, slope: float = 1.0) -> Act:
    return Act(buf, 0, buf.shape[1], scale, shift, slope)


# ---------------------------------------------------------------------------
# FFT family
# ---------------------------------------------------------------------------
def fft_workspace(planes: int, h: int, w: int, device) -> torch.Tensor:
    nbytes = lib().query("san_fft_workspace_bytes", planes, h, w)
    return GLOBAL_ARENA.get("fft_ws", (nbytes // 4,), device)


def fft2c(x: torch.Tensor, inverse: bool = False, colmask_in: Optional[torch.Tensor] = None,
          colmask_out: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Ortho (i)fft2 of a complex64 [N, C, H, W] tensor, optional [W] column masks."""
    xr = _creal(x, "x")
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty_like(x)
    ws = fft_workspace(n * c, h, w, x.device)
    lib().call("san_fft2", _p(xr), _p(_creal(out, "out")), n * c, h, w, int(inverse), _p(colmask_in), _p(colmask_out), 0,
               _p(ws), ws.numel() * 4, _stream())
    return out


def ifft2c_planar(x: torch.Tensor, colmask_in: Optional[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
    """ifft2(x * colmask) written planar into real ``out`` [N*C, ctot>=2, H, W]."""
    xr = _creal(x, "x")
    n, c, h, w = x.shape
    _chk(out, name="out")
    assert out.shape[0] == n * c and out.shape[2:] == (h, w)
    ws = fft_workspace(n * c, h, w, x.device)
    lib().call("san_fft2", _p(xr), _p(out), n * c, h, w, 1, _p(colmask_in), _p(None), int(out.shape[1]),
               _p(ws), ws.numel() * 4, _stream())
    return out


def sens_reduce(k: torch.Tensor, sens: torch.Tensor, out: torch.Tensor, cols: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum_c ifft2(k)*conj(sens) -> channels 0,1 of real ``out`` [N, ctot, H, W].  ``cols``: the inverse
    column transform of k if the previous sens_expand_dc already produced it (then only the row pass runs)."""
    n, c, h, w = k.shape
    _chk(out, name="out")
    ws = fft_workspace(n * c, h, w, k.device)
    src = k if cols is None else cols
    args = (_p(_creal(src, "k")), _p(_creal(sens, "sens")), _p(out), int(out.shape[1]), n, c, h, w, _p(ws), ws.numel() * 4,
            _stream())
    fn = "san_sens_reduce" if cols is None else "san_sens_reduce_from_cols"
    # algorithmic bytes: read k and S (C planes each), write m (1 plane); E = H*W*8
    # (its own family: one launch per forward pass, not a cascade boundary -- as "fft_dc" it was always the first, sampled, launch)
    _timed("fft_sens", float((2 * c + 1) * n * h * w * 8), "B", lambda: lib().call(fn, *args))
    return out


def sens_expand_dc(r_planar: torch.Tensor, sens: torch.Tensor, k: torch.Tensor, k0: torch.Tensor,
                   mask: torch.Tensor, dc_w: torch.Tensor, k_out: torch.Tensor,
                   next_cols: Optional[torch.Tensor] = None) -> torch.Tensor:
    """k_out = k - dc_w*mask*(k - k0) - fft2(r*S).  ``next_cols`` (complex, like k_out): also receives the
    inverse column transform of k_out, the first pass of the next cascade's sens_reduce / of ifft2_rss."""
    n, c, h, w = k.shape
    _chk(r_planar, name="r_planar")
    assert r_planar.shape == (n, 2, h, w)
    ws = fft_workspace(n * c, h, w, k.device)
    head = (_p(r_planar), _p(_creal(sens, "sens")), _p(_creal(k, "k")), _p(_creal(k0, "k0")),
            _p(_chk(mask, name="mask")), _p(_chk(dc_w, name="dc_w")), _p(_creal(k_out, "k_out")))
    tail = (n, c, h, w, _p(ws), ws.numel() * 4, _stream())
    if next_cols is None:
        fn, args = "san_sens_expand_dc", head + tail
    else:
        fn, args = "san_sens_expand_dc_next", head + (_p(_creal(next_cols, "next_cols")),) + tail
    # algorithmic bytes: read r (1 plane), S, k, k0 (C planes each), write k' (C planes)
    _timed("fft_sens", float((4 * c + 1) * n * h * w * 8), "B", lambda: lib().call(fn, *args))
    return k_out


def fft_cols(x: torch.Tensor, inverse: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Ortho (i)fft along H only of complex [N, C, H, W] (k0x = ifft_y(k0): the data term of dc_rows)."""
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty_like(x)
    lib().call("san_fft_cols", _p(_creal(x, "x")), _p(_creal(out, "out")), n * c, h, w, int(inverse), _stream())
    return out


def dc_rows(x: torch.Tensor, sens: torch.Tensor, k0x: Optional[torch.Tensor], mask: torch.Tensor, dc_w: torch.Tensor,
            r_planar: Optional[torch.Tensor], x_out: Optional[torch.Tensor], m_out: Optional[torch.Tensor],
            dk_out: Optional[torch.Tensor] = None, m_stats: Optional[torch.Tensor] = None) -> None:
    """One cascade boundary in the image domain (san_dc_rows): x_out = x - dc_w ifft_x(mask (fft_x(x) - k0x)) - r S and
    m_out[:, 0:2] = sum_c conj(S_c) x_out_c (planar, into a [N, ctot, H, W] buffer).  ``m_stats`` (dc_rows_stat_part): the launch
    also emits the (count, mean, M2) records of m_out's two planes -- the next NormUnet's statistics (san_dc_rows_stats, round 6)."""
    n, c, h, w = x.shape
    head = (_p(_creal(x, "x")), _p(_creal(sens, "sens")), _p(None if k0x is None else _creal(k0x, "k0x")),
            _p(_chk(mask, name="mask")), _p(_chk(dc_w, name="dc_w")), _p(None if r_planar is None else _chk(r_planar, name="r")),
            _p(None if x_out is None else _creal(x_out, "x_out")), _p(None if m_out is None else _chk(m_out, name="m_out")),
            int(m_out.shape[1]) if m_out is not None else 2, _p(None if dk_out is None else _creal(dk_out, "dk_out")))
    if m_stats is not None:
        assert m_out is not None and tuple(m_stats.shape) == (n, 2, lib().query("san_dc_rows_stat_tiles", n, c, h, w), 3)
        fn, args = "san_dc_rows_stats", head + (_p(_chk(m_stats, name="m_stats")), n, c, h, w, _stream())
    else:
        fn, args = "san_dc_rows", head + (_p(None), _p(None), 0, n, c, h, w, _stream())
    # SURVEY 8(d) bytes of one cascade's FFT + DC work: (6C + 2) planes of H*W*8 B per slice; this kernel itself moves
    # (4C + 2) (+ C for dk_out): the column passes of the two 2-D transforms are gone
    _timed("fft_dc", float((6 * c + 2) * n * h * w * 8), "B", lambda: lib().call(fn, *args),
           float((4 * c + 2 + (c if dk_out is not None else 0)) * n * h * w * 8))


def dc_rows_stat_part(n: int, c: int, h: int, w: int, device, arena: Arena = GLOBAL_ARENA, tag: str = "") -> Optional[torch.Tensor]:
    """The records buffer for ``dc_rows(..., m_stats=)``, or None where the shape's kernel does not emit statistics."""
    tiles = lib().query("san_dc_rows_stat_tiles", n, c, h, w)
    return arena.get("dcstat" + tag, (n, 2, tiles, 3), device) if tiles else None


def dc_rows_bwd(g: torch.Tensor, sens: torch.Tensor, mask: torch.Tensor, dc_w: torch.Tensor, g_out: torch.Tensor,
                h_out: torch.Tensor, dk: torch.Tensor, dcw_grad: Optional[torch.Tensor] = None,
                defer_dcw: bool = False) -> Optional[torch.Tensor]:
    """Backward form: g_out = g - dc_w ifft_x(mask fft_x(g)); h_out = -sum_c conj(S_c) g_c (planar: the gradient wrt the
    regulariser output); dL/d(dc_w) = - (fixed-order sum of the per-workgroup partials): added to ``dcw_grad`` (a 1-element
    fp32 tensor) on the device, or returned as a 0-d tensor."""
    n, c, h, w = g.shape
    part = GLOBAL_ARENA.get("dcw_part", (lib().query("san_dc_rows_partials", n, c, h, w),), g.device)
    bargs = (_p(_creal(g, "g")), _p(_creal(sens, "sens")), _p(None), _p(_chk(mask, name="mask")),
             _p(_chk(dc_w, name="dc_w")), _p(None), _p(_creal(g_out, "g_out")), _p(_chk(h_out, name="h_out")),
             int(h_out.shape[1]), _p(None), _p(_creal(dk, "dk")), _p(part), 1, n, c, h, w, _stream())
    # the adjoint launch does the forward boundary's work on the gradient: same SURVEY 8(d) count, its own family
    _timed("fft_dc_bwd", float((6 * c + 2) * n * h * w * 8), "B", lambda: lib().call("san_dc_rows", *bargs),
           float((5 * c + 2) * n * h * w * 8))
    if defer_dcw:
        return part                             # the caller adds -sum(part) to dc_weight's gradient later (normunet_bwd_tail)
    if dcw_grad is not None:
        partials_add(part, -1.0, dcw_grad)
        return None
    return -(part.double().sum()).float()


def partials_add(part: torch.Tensor, scale: float, dst: torch.Tensor) -> None:
    """dst[0] += scale * sum(part): fixed-order double accumulation on the device (san_partials_add)."""
    assert dst.numel() == 1 and dst.dtype == torch.float32 and dst.is_contiguous()
    lib().call("san_partials_add", _p(part), int(part.numel()), float(scale), _p(dst), _stream())


def sens_grad_prop(gS: Optional[torch.Tensor], r_planar: torch.Tensor, t1: torch.Tensor, x: torch.Tensor, gm_planar: torch.Tensor,
                   gd: torch.Tensor, sens: torch.Tensor) -> None:
    """gd += gm * S (in place) and, if gS is given, the sensitivity-map gradient of one cascade (san_sens_grad_prop)."""
    n, c, h, w = gd.shape
    lib().call("san_sens_grad_prop", _p(None if gS is None else _creal(gS, "gS")), _p(_chk(r_planar, name="r")),
               _p(_creal(t1, "t1")), _p(_creal(x, "x")), _p(_chk(gm_planar, name="gm")), -1.0, _p(_creal(gd, "gd")),
               _p(_creal(sens, "sens")), n, c, h * w, _stream())


def ifft2_rss(k: torch.Tensor, out: Optional[torch.Tensor] = None, cols: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, c, h, w = k.shape
    if out is None:
        out = torch.empty((n, 1, h, w), device=k.device, dtype=torch.float32)
    ws = fft_workspace(n * c, h, w, k.device)
    src, fn = (k, "san_ifft2_rss") if cols is None else (cols, "san_ifft2_rss_from_cols")
    lib().call(fn, _p(_creal(src, "k")), _p(_chk(out, name="out")), n, c, h, w, _p(ws), ws.numel() * 4, _stream())
    return out


def sens_normalize(est_planar: torch.Tensor, n: int, c: int) -> torch.Tensor:
    _chk(est_planar, name="est_planar")
    h, w = est_planar.shape[2:]
    sens = torch.empty((n, c, h, w), device=est_planar.device, dtype=torch.complex64)
    lib().call("san_sens_normalize", _p(est_planar), _p(torch.view_as_real(sens)), n, c, h, w, _stream())
    return sens


def rss(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, c = x.shape[:2]
    hw = x.shape[2] * x.shape[3]
    if out is None:
        out = torch.empty((n, 1) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
    if torch.is_complex(x):
        lib().call("san_rss", _p(_creal(x, "x")), _p(out), n, c, hw, 1, _stream())
    else:
        lib().call("san_rss", _p(_chk(x, name="x")), _p(out), n, c, hw, 0, _stream())
    return out


def cabs(x: torch.Tensor) -> torch.Tensor:
    """|x| per element of a complex [N,C,H,W] tensor (rss over a single channel)."""
    n, c, h, w = x.shape
    return rss(x.contiguous().reshape(n * c, 1, h, w)).reshape(n, c, h, w)


# ---------------------------------------------------------------------------
# conv / norm stack
# ---------------------------------------------------------------------------
# In-place parameter updates by san_adamw_step do not touch tensor._version: the packing caches
# below also key on this counter, which the fused optimiser bumps after every step.
WEIGHT_EPOCH = [0]


def bump_weight_epoch() -> None:
    WEIGHT_EPOCH[0] += 1


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float,
               beta2: float, eps: float, weight_decay: float, step: int, grad_scale: float = 1.0) -> None:
    """One fused AdamW step over flat fp32 buffers (see san_adamw_step)."""
    for t, name in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, name=name)
    assert p.numel() == g.numel() == m.numel() == v.numel()
    lib().call("san_adamw_step", _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay,
               int(step), grad_scale, _stream())


def adamw_step_dev(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float,
                   eps: float, weight_decay: float, step_dev: torch.Tensor, grad_scale: float = 1.0,
                   hyper_dev: Optional[torch.Tensor] = None) -> None:
    """adamw_step with the step count in device memory (int64 [1]): capture-safe (san_adamw_step_hyper).  hyper_dev (fp32 [3]
    = lr, weight_decay, grad_scale on the device) overrides the scalar arguments: a captured step follows a schedule."""
    for t, name in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, name=name)
    _chk(step_dev, torch.int64, "step_dev")
    if hyper_dev is not None:
        assert _chk(hyper_dev, name="hyper_dev").numel() == 3
    lib().call("san_adamw_step_hyper", _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay,
               _p(step_dev), grad_scale, _p(hyper_dev), _stream())


class _PackRegistry:
    """Packed copies of conv weights (the layouts the MFMA kernels read).  Inference packs a weight
    once; in training every weight changes every step, so ALL registered (weight, mode) pairs are
    re-packed by one batched launch (san_conv_pack_batch) the first time any of them is requested
    after an optimiser step, instead of ~650 tiny launches per step.  mode 0: Conv2d forward,
    1: ConvTranspose2d 2x2, 2: Conv2d data gradient."""

    def __init__(self):
        self.jobs = {}          # (id(w), mode) -> job dict
        self.table = None       # device int64 [njobs, 8]
        self.order = []
        self.epoch = -1

    def _prune(self):
        dead = [k for k, j in self.jobs.items() if j["wref"]() is None]
        for k in dead:
            del self.jobs[k]
        if dead:
            self.table = None

    def _alloc(self, w: torch.Tensor, mode: int):
        """(dims, packed buffer) for a new job."""
        if mode == 1:
            cin, cout, ks = w.shape[0], w.shape[1], w.shape[2]
            nfl = lib().query("san_conv_packed_floats", cout, cin, ks)
        elif mode == 2:
            cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
            nfl = lib().query("san_conv_packed_floats", cin, cout, ks)
        else:
            cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
            nfl = lib().query("san_conv_packed_floats", cout, cin, ks)
        return (cout, cin, ks), torch.empty(nfl, device=w.device, dtype=torch.float32)

    def _register(self, w: torch.Tensor, mode: int):
        _chk(w, name="weight")
        dims, packed = self._alloc(w, mode)
        if len(self.jobs) >= 4096:
            self._prune()
        job = {"wref": weakref.ref(w), "ptr": w.data_ptr(), "mode": mode, "dims": dims, "version": -1,
               "packed": packed, "device": w.device}
        self.jobs[(id(w), mode)] = job
        self.table = None
        return job

    def _fill_job(self, row, j):
        cout, cin, ks = j["dims"]
        lib().call("san_conv_pack_job", ctypes.c_void_p(ctypes.addressof(row)), ctypes.c_void_p(j["ptr"]),
                   _p(j["packed"]), cout, cin, ks, j["mode"])

    def _batch(self):
        lib().call("san_conv_pack_batch", _p(self.table), len(self.order), _stream())

    def ensure_table(self, device):
        """Build (upload) the device job table if it is stale.  Called before a hipGraph capture: the upload is a
        host-to-device copy, which a capturing stream does not allow."""
        if self.table is not None and any(j["wref"]() is None for j in self.order):
            self.table = None                   # a registered weight was freed: never read through its old pointer
        if self.table is None or self.table.device != device:
            self._prune()
            self.order = [j for j in self.jobs.values() if j["device"] == device]
            host = torch.zeros((len(self.order), 8), dtype=torch.int64)
            row = (ctypes.c_longlong * 8)()
            for i, j in enumerate(self.order):
                self._fill_job(row, j)
                host[i] = torch.tensor(list(row), dtype=torch.int64)
            self.table = host.to(device)

    def _run(self, device):
        """Re-pack every live job on `device` in one launch."""
        self.ensure_table(device)
        if self.order:
            if _lib.KEEP is not None:
                # inside a recording: the recorded launch reads THIS job table and writes THESE packed images at every replay.  A
                # table rebuilt during the recorded step itself (a registered weight of some other model died just then) used to
                # be held by the registry only -- after the next rebuild a replay read whatever the allocator had put there
                # (typically the NEW table, with the OLD job count: some images silently not re-packed, weights one step stale)
                _lib.KEEP.append(self.table)
                _lib.KEEP.append([j["packed"] for j in self.order])
            self._batch()
        for j in self.order:
            w = j["wref"]()
            j["version"] = w._version if w is not None else -1
        self.epoch = WEIGHT_EPOCH[0]

    def _pack_one(self, job, w):
        cout, cin, ks = job["dims"]
        if job["mode"] == 1:
            lib().call("san_conv_pack_weights", _p(w.detach()), _p(job["packed"]), cout, cin, ks, 1, _stream())
        elif job["mode"] == 2:
            lib().call("san_conv_pack_weights_dgrad", _p(w.detach()), _p(job["packed"]), cout, cin, ks, _stream())
        else:
            lib().call("san_conv_pack_weights_fwd", _p(w.detach()), _p(job["packed"]), cout, cin, ks, _stream())
        job["version"] = w._version

    def get(self, w: torch.Tensor, mode: int) -> torch.Tensor:
        job = self.jobs.get((id(w), mode))
        if job is None or job["wref"]() is not w or job["ptr"] != w.data_ptr():
            job = self._register(w, mode)
        if self.epoch != WEIGHT_EPOCH[0]:
            self._run(w.device)                 # an optimiser step happened: everything is stale, one launch
        if job["version"] != w._version:
            self._pack_one(job, w)              # new weight, or modified in place by torch (load_state_dict, tests)
        return job["packed"]


PACKS = _PackRegistry()


def _scale_tail_one(buf: torch.Tensor) -> torch.Tensor:
    """The 16 bytes behind a packed image hold the tensor's power-of-two scale {S_w, 1 / S_w} (fp8 format always, fp16 format when
    f16_weight_scale is on): {1, 1} until a scale pass writes them."""
    buf[-16:-8].view(torch.float32).fill_(1.0)
    return buf


F16_WSCALE = [os.environ.get("SAN_F16_WSCALE", "0") == "1"]


def f16_weight_scale(on: bool) -> bool:
    """Switch the per-tensor power-of-two scale of the fp16-format weight images (san_conv_f16_wscale_enable, round 6) on or off;
    returns the previous setting.  Every registered image is re-packed (scaled, or unscaled with its scale words back at 1)."""
    prev = bool(lib().query("san_conv_f16_wscale_enable", 1 if on else 0))
    F16_WSCALE[0] = bool(on)
    for j in PACKS16.jobs.values():
        if j["mode"] & 16:
            _scale_tail_one(j["packed"])
            j["version"] = -1
    PACKS16.table = None
    PACKS16.epoch = -1
    return prev


class _PackRegistryBf16(_PackRegistry):
    """The same bookkeeping for the bf16x3 kernels' weight images (csrc/san_conv_bf16.hip): mode 0 forward,
    mode 2 data gradient; dims = (cout, cin) of the convolution that will run."""

    def _alloc(self, w: torch.Tensor, mode: int):
        with _lib.untracked():                          # (a one-time zero fill: not part of a recorded step)
            return self._alloc_zeroed(w, mode)

    def _alloc_zeroed(self, w: torch.Tensor, mode: int):
        mode &= 15                                      # (+16 = two fp16 parts, +32 = one fp8 part instead of bf16 parts: same image size)
        if mode == 3:                                   # ConvTranspose2d [Cin, Cout, 2, 2] as a 1x1 conv to 4 Cout channels
            return (4 * w.shape[1], w.shape[0], 1), _scale_tail_one(torch.zeros(
                lib().query("san_conv_bf16x3_packed_bytes_ks", 4 * w.shape[1], w.shape[0], 1), device=w.device, dtype=torch.uint8))
        if mode == 2:
            cout, cin = w.shape[1], w.shape[0]          # the data-gradient conv maps forward cout -> forward cin
        else:
            cout, cin = w.shape[0], w.shape[1]
        ks = int(w.shape[2])
        nbytes = lib().query("san_conv_bf16x3_packed_bytes_ks", cout, cin, ks)
        return (cout, cin, ks), _scale_tail_one(torch.zeros(nbytes, device=w.device, dtype=torch.uint8))     # (zero-filled: the batched pack skips constant-zero units)

    def _fill_job(self, row, j):
        cout, cin, ks = j["dims"]
        lib().call("san_conv_bf16x3_pack_job_ks", ctypes.c_void_p(ctypes.addressof(row)), ctypes.c_void_p(j["ptr"]),
                   _p(j["packed"]), cout, cin, self._cmode(j["mode"]), ks)

    @staticmethod
    def _cmode(mode: int) -> int:
        """registry mode -> library mode: 3 (transposed) packs like a data gradient (2); bits 4 (fp16 parts) / 5 (fp8) pass through"""
        return (2 if (mode & 15) == 3 else (mode & 15)) | (mode & 48)

    @staticmethod
    def _blocks(j) -> int:
        """Workgroups for one job: a power of two from 1 (a few hundred weights) to 64 (a 288 x 288 x 9 layer), ~8 K packed bytes each."""
        b, want = 1, max(1, j["packed"].numel() // 8192)
        while b < 64 and b < want:
            b *= 2
        return b

    def ensure_table(self, device):
        """As the base class, with the jobs sorted by size class: ``_batch`` packs each class with a grid that fits it (one launch
        of 64 workgroups per job spent 0.3 ms per step dispatching ~42,000 idle workgroups)."""
        stale = self.table is None or self.table.device != device or any(j["wref"]() is None for j in self.order)
        if stale:
            self._prune()
            self.jobs = {k: v for k, v in sorted(self.jobs.items(), key=lambda kv: -self._blocks(kv[1]))}
            self.table = None
        super().ensure_table(device)
        if stale:
            self.classes = []                       # (first job, count, workgroups per job, any fp8 image)
            for i, j in enumerate(self.order):
                b, f8 = self._blocks(j), 1 if ((j["mode"] & 32) or ((j["mode"] & 16) and F16_WSCALE[0])) else 0     # (images with a per-tensor scale: the wscale pass)
                if self.classes and self.classes[-1][2] == b:
                    first, cnt, _, f = self.classes[-1]
                    self.classes[-1] = (first, cnt + 1, b, f | f8)
                else:
                    self.classes.append((i, 1, b, f8))

    def _batch(self):
        base = self.table.data_ptr()
        for first, cnt, blocks, f8 in self.classes:
            lib().call("san_conv_bf16x3_pack_batch_grid", base + first * 64, cnt, blocks, f8, _stream())

    def _pack_one(self, job, w):
        cout, cin, ks = job["dims"]
        lib().call("san_conv_bf16x3_pack_ks", _p(w.detach()), _p(job["packed"]), cout, cin, self._cmode(job["mode"]),
                   ks, _stream())
        job["version"] = w._version


PACKS16 = _PackRegistryBf16()
# Forward convolutions of the fp32-equivalent mode run on TWO fp16 parts per operand (three products instead of six bf16
# ones; csrc/san_conv_bf16.hip "f16x2"): their operands are normalised activations and weights, for which fp16's range
# is ample.  Gradients (1e-7-sized) keep the three-part bf16 split.  SAN_NO_F16X2=1 switches it off.
F16_FWD = [os.environ.get("SAN_NO_F16X2", "0") != "1"]
_CONV_NP = [3]
# "fp8" mode (BASELINE config 5): FORWARD convolutions on one OCP e4m3 part per operand (per-tensor power-of-two weight scale,
# activations x 8: csrc/san_conv_bf16.hip "fp8"); data and weight gradients on plain bf16; FFT / DC / norms / losses fp32.
_FP8_FWD = [False]


def _fwd_fmt() -> int:
    if _FP8_FWD[0] and _CONV_NP[0] == 1:
        return 32
    return 16 if (F16_FWD[0] and _CONV_NP[0] == 3) else 0


# Gradients on two fp16 parts need a per-tensor power-of-two scale: the kernels that WRITE a dy tensor (san_act_bwd*_amax)
# keep its largest magnitude in one amax record of this pool (64 lines, integer atomic max of the float bits: deterministic), the data /
# weight gradient kernels that READ it derive the scale from that slot.  CSModel.backward() resets the pool once per step.
F16_BWD = [os.environ.get("SAN_NO_F16X2_BWD", os.environ.get("SAN_NO_F16X2", "0")) != "1"]


class AmaxPool:
    """Amax records (san_hip.h: san_amax_record_words() int32 words each, 64 atomic lines) handed out one per dy tensor per
    step; ``reset`` (once per step) zeroes the records the previous step used."""
    SLOTS = 4096

    def __init__(self):
        self.buf = {}
        self.views = {}
        self.idx = 0
        self.words = None

    def _words(self) -> int:
        if self.words is None:
            self.words = int(lib().query("san_amax_record_words"))
        return self.words

    def reset(self, device=None) -> None:
        """Zero the records handed out since the last reset (one memset) and start over.  ``device`` may be given with or
        without an index ('cuda' == the current device)."""
        used = self.idx * self._words()
        want = None if device is None else _dev_key(device)
        for key, t in self.buf.items():
            if used and (want is None or key == want):
                _lib.rec(t[:used].zero_)
        self.idx = 0

    def next(self, device) -> Optional[torch.Tensor]:
        if not (F16_BWD[0] and _CONV_NP[0] == 3):
            return None
        key = _dev_key(device)
        t = self.buf.get(key)
        w = self._words()
        if t is None:
            t = self.buf[key] = torch.zeros(self.SLOTS * w, dtype=torch.int32, device=device)
        if self.idx >= self.SLOTS:
            # wrap-around without a per-step reset (a caller driving module.backward in a loop): side-stream weight
            # gradients may still be reading the records about to be zeroed -- join them first
            side = _WG["stream"]
            if side is not None:
                _flush_pending()
                torch.cuda.current_stream().wait_stream(side)
            self.reset(device)
        self.idx += 1
        slots = self.views.get(key)
        if slots is None:
            slots = self.views[key] = [None] * self.SLOTS
        v = slots[self.idx - 1]
        if v is None:
            v = slots[self.idx - 1] = t[(self.idx - 1) * w:self.idx * w]       # (a view per slot, made once: not per step)
        return v


def amax_record(value: torch.Tensor) -> torch.Tensor:
    """An amax record holding ``value`` (a non-negative float32 scalar tensor on the device) in every line (tests, benches)."""
    w = int(lib().query("san_amax_record_words"))
    return value.detach().reshape(1).float().view(torch.int32).repeat(w).contiguous()


def amax_value(rec: torch.Tensor) -> float:
    """The maximum an amax record holds."""
    return float(rec.view(torch.float32).max().item())


AMAX = AmaxPool()

# Arithmetic of the matrix-core convolutions / weight gradients (san_set_conv_precision).  "bf16x3" is the default and the
# only mode held to the 1e-4 parity bar; "bf16x2" / "bf16" are the narrow-precision modes (PSNR-judged).
CONV_PRECISIONS = {"bf16x3": 3, "fp32": 3, "bf16x2": 2, "bf16": 1, "fp8": 1}


def set_conv_precision(mode: str) -> str:
    """Select the operand parts of every bf16 matrix-core kernel (process-wide); returns the previous mode's name."""
    if mode not in CONV_PRECISIONS:
        raise ValueError(f"conv precision {mode!r}: choose from {sorted(CONV_PRECISIONS)}")
    prev = {3: "bf16x3", 2: "bf16x2", 1: "fp8" if _FP8_FWD[0] else "bf16"}[lib().query("san_get_conv_precision")]
    lib().call("san_set_conv_precision", CONV_PRECISIONS[mode])
    _CONV_NP[0] = CONV_PRECISIONS[mode]
    _FP8_FWD[0] = mode == "fp8"
    return prev


def current_precision() -> str:
    """The mode set_conv_precision last selected."""
    return {3: "bf16x3", 2: "bf16x2", 1: "fp8" if _FP8_FWD[0] else "bf16"}[_CONV_NP[0]]


def conv_direct(on: bool) -> bool:
    """Switch the direct fp32 kernel for layers with <= 4 channels on one side (csrc/san_conv_mfma.hip, round 5) on or off;
    returns the previous setting.  Off: those layers run on the outer-product kernel as before (A/B runs, tests).  The
    statistics-tile geometry differs between the two, so the memoised geometry queries are dropped."""
    prev = bool(lib().query("san_conv_direct_enable", 1 if on else 0))
    lib()._memo.clear()
    return prev


def conv1x1_gemm(on: bool) -> bool:
    """Switch the one-stage GEMM form of the 1x1 / transposed convolutions (csrc/san_conv1x1.hip, round 5) on or off; returns the
    previous setting.  Off: the tiled kernel's KS = 1 form, as before (A/B runs, tests)."""
    return bool(lib().query("san_conv1x1_gemm_enable", 1 if on else 0))


class conv_precision:
    """``with ops.conv_precision("bf16"): ...`` -- the mode inside the block, the previous one afterwards."""

    def __init__(self, mode: str):
        self.mode = mode

    def __enter__(self):
        self.prev = set_conv_precision(self.mode)
        return self

    def __exit__(self, *exc):
        set_conv_precision(self.prev)
        return False
USE_BF16X3 = [os.environ.get("SAN_NO_BF16X3", "0") != "1"]


# Data gradients to 2 / 3 channels (a cascade's input convolution) on the persistent matrix-core kernel instead of the direct fp32 one
# (round 6, san_conv_stream_eligible; SAN_STREAM_SMALL_COUT=0: off -- read once, by the library too)
STREAM_SMALL_COUT = [os.environ.get("SAN_STREAM_SMALL_COUT", "1") != "0"]


def bf16x3_eligible(cin: int, cout: int, h: int, w: int, ks: int) -> bool:
    if not USE_BF16X3[0]:
        return False
    if ks == 1:
        return bool(lib().query("san_conv1x1_bf16x3_eligible", cin, cout, h, w))
    return bool(lib().query("san_conv_bf16x3_eligible", cin, cout, h, w, ks))


def packed_weight(w: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """Packed copy of a Conv2d (or, transposed, ConvTranspose2d 2x2) weight for the MFMA kernels."""
    return PACKS.get(w, 1 if transposed else 0)


def _bf16x3_launch(ks: int, bargs, n: int, h: int, w: int, cin: int, cout: int, device, arena: "Arena", fin=None) -> bool:
    """bargs = the arguments of san_conv2d_bf16x3_fwd (stream last).  3x3 layers that the library wants to split over K
    (deep K, few tiles: san_conv_bf16x3_ws_bytes > 0) get the scratch for the partial outputs.  fin = (scale, shift, coff,
    eps) of a following InstanceNorm: a split launch finalises the lazy affine in its reduction pass (returns True)."""
    if ks != 3:
        lib().call("san_conv1x1_bf16x3_fwd", *bargs)
        return False
    nbytes = lib().query("san_conv_bf16x3_ws_bytes", n, h, w, cin, cout, 3)
    if nbytes == 0:
        lib().call("san_conv2d_bf16x3_fwd", *bargs)
        return False
    ws = arena.scratch("b16_splitk", nbytes, device)
    if fin is not None and bargs[13] is not None:
        scale, shift, coff, eps = fin
        done = ctypes.c_int(0)
        if _lib.KEEP is not None:
            _lib.KEEP.append(done)              # its address is part of the recorded call
        lib().call("san_conv2d_bf16x3_fwd_ws_in", *bargs[:-1], _p(ws), nbytes, _p(scale), _p(shift), int(scale.shape[1]), int(coff),
                   float(eps), ctypes.c_void_p(ctypes.addressof(done)), bargs[-1])
        return bool(done.value)
    lib().call("san_conv2d_bf16x3_fwd_ws", *bargs[:-1], _p(ws), nbytes, bargs[-1])
    return False


def conv2d(x: Act, weight: torch.Tensor, bias: Optional[torch.Tensor], y: Act, stats: bool = False,
           out_scale: Optional[torch.Tensor] = None, out_shift: Optional[torch.Tensor] = None,
           arena: Arena = GLOBAL_ARENA, tag: str = "", grad_input: bool = False,
           instance_norm_eps: Optional[float] = None) -> Optional[torch.Tensor]:
    """y.buf[:, y.coff:y.coff+cout] = conv(T(x)) (+bias).  Returns the per-tile
    statistics partials [N, cout, tiles, 3] when ``stats``.  ``grad_input``: x is a gradient (arbitrary magnitude):
    keep the bf16 operand split, whose exponent range is fp32's.  ``instance_norm_eps``: the layer is followed by
    InstanceNorm2d on y (y.scale / y.shift): where the launch can finalise that affine itself (split-K layers) it does and
    None is returned -- the caller runs norm_finalize only on a returned tensor."""
    cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[2]
    assert cin == x.c and cout == y.c, (cin, x.c, cout, y.c)
    assert x.buf.shape[2:] == y.buf.shape[2:]
    n, h, w = x.n, x.h, x.w
    part = None
    # (round 6) a 3x3 convolution TO 2 / 3 channels (the alignment net's last layer, unet.py:108-111) on the persistent matrix-core
    # kernel with those channels as a partial block alone, where its shape rules hold -- the one-tile kernel does not take such a layer
    small = (STREAM_SMALL_COUT[0] and ks == 3 and cout in (2, 3) and cin >= 16 and out_scale is None and not stats and not grad_input
             and USE_BF16X3[0] and _fwd_fmt() == 16 and y.buf.data_ptr() % 16 == 0
             and lib().query("san_conv_stream_eligible", n, h, w, cin, cout, x.ctot))
    if out_scale is None and (small or bf16x3_eligible(cin, cout, h, w, ks)):
        # bf16 matrix cores, operands split in three (fp32-level accuracy), csrc/san_conv_bf16.hip
        fmt = (16 if (x.amax is not None and _CONV_NP[0] == 3) else 0) if grad_input else _fwd_fmt()
        wp = PACKS16.get(weight, fmt)
        if grad_input and fmt:
            # a gradient through a plain convolution (the transposed convolution's data gradient): fp16 parts, scaled by its maximum
            nbytes = lib().query("san_conv_bf16x3_ws_bytes", n, h, w, cin, cout, 3) if ks == 3 else 0
            ws = arena.scratch("b16_splitk", nbytes, x.buf.device) if nbytes else None
            gargs = (_p(x.buf), x.ctot, x.coff, cin, _p(wp), _p(y.buf), y.ctot, y.coff, cout, _p(x.amax), n, h, w, ks, _p(ws), nbytes,
                     _stream())
            _timed("conv3x3_bf16x3" if ks == 3 else "conv1x1_bf16x3", 2.0 * n * h * w * cout * cin * ks * ks, "FLOP",
                   lambda: lib().call("san_conv_bf16x3_dgrad_amax", *gargs), _conv_abytes(n, h, w, cin, cout, ks), 3)
            return None
        if stats:
            tiles = (lib().query("san_conv3x3_bf16x3_stat_tiles", n, h, w, cin, cout, int(fmt == 16)) if ks == 3
                     else lib().query("san_conv_bf16x3_stat_tiles", n, h, w))
            part = arena.get("part" + tag, (n, cout, tiles, 3), x.buf.device)
        bargs = (_p(x.buf), x.ctot, x.coff, cin, _p(x.scale), _p(x.shift), float(x.slope), _p(wp), _p(bias), _p(y.buf),
                 y.ctot, y.coff, cout, _p(part), n, h, w, _stream())
        fin = (y.scale, y.shift, y.coff, instance_norm_eps) if (instance_norm_eps is not None and stats and y.scale is not None) else None
        done = [False]

        def _go():
            done[0] = _bf16x3_launch(ks, bargs, n, h, w, cin, cout, x.buf.device, arena, fin)

        # (the 2- / 3-channel layers of the persistent kernel are their own family: a sixth of a channel block's work is algorithmic)
        fam = "conv3x3_few_cout" if small else ("conv3x3_bf16x3" if ks == 3 else "conv1x1_bf16x3")
        _timed(fam, 2.0 * n * h * w * cout * cin * ks * ks, "FLOP", _go, _conv_abytes(n, h, w, cin, cout, ks), _products(fmt != 0))
        return None if done[0] else part
    wp = packed_weight(weight)
    if stats:
        tiles = lib().query("san_conv_stat_tiles", n, h, w, cin, cout, ks)
        part = arena.get("part" + tag, (n, cout, tiles, 3), x.buf.device)
    args = (_p(x.buf), x.ctot, x.coff, cin, _p(x.scale), _p(x.shift), float(x.slope), _p(wp), _p(bias), _p(y.buf),
            y.ctot, y.coff, cout, _p(out_scale), _p(out_shift), _p(part), n, h, w, ks, _stream())
    _timed("conv3x3" if ks == 3 else "conv1x1", 2.0 * n * h * w * cout * cin * ks * ks, "FLOP",
           lambda: lib().call("san_conv2d_fwd", *args), _conv_abytes(n, h, w, cin, cout, ks))
    return part


def tconv2x2(x: Act, weight: torch.Tensor, y: Act, stats: bool = False, arena: Arena = GLOBAL_ARENA,
             tag: str = "") -> Optional[torch.Tensor]:
    cin, cout = weight.shape[0], weight.shape[1]
    assert cin == x.c and cout == y.c
    assert y.h == 2 * x.h and y.w == 2 * x.w
    part = None
    if USE_BF16X3[0] and lib().query("san_tconv2x2_bf16x3_eligible", cin, cout, x.h, x.w) and y.buf.data_ptr() % 8 == 0:
        wp = PACKS16.get(weight, 3 + _fwd_fmt())
        if stats:
            tiles = 4 * lib().query("san_conv_bf16x3_stat_tiles", x.n, x.h, x.w)
            part = arena.get("tpart" + tag, (x.n, cout, tiles, 3), x.buf.device)
        bargs = (_p(x.buf), x.ctot, x.coff, cin, _p(x.scale), _p(x.shift), float(x.slope), _p(wp), _p(y.buf), y.ctot, y.coff,
                 cout, _p(part), x.n, x.h, x.w, _stream())
        _timed("tconv2x2_bf16x3", 2.0 * x.n * x.h * x.w * 4 * cout * cin, "FLOP", lambda: lib().call("san_tconv2x2_bf16x3_fwd", *bargs))
        return part
    wp = packed_weight(weight, transposed=True)
    if stats:
        tiles = lib().query("san_tconv_stat_tiles", x.n, x.h, x.w, cout)
        part = arena.get("tpart" + tag, (x.n, cout, tiles, 3), x.buf.device)
    args = (_p(x.buf), x.ctot, x.coff, cin, _p(x.scale), _p(x.shift), float(x.slope), _p(wp), _p(y.buf), y.ctot, y.coff,
            cout, _p(part), x.n, x.h, x.w, _stream())
    _timed("tconv2x2", 2.0 * x.n * x.h * x.w * 4 * cout * cin, "FLOP", lambda: lib().call("san_tconv2x2_fwd", *args))
    return part


def norm_finalize(part: torch.Tensor, mode: int, eps: float, scale: torch.Tensor, shift: torch.Tensor, coff: int,
                  gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
                  aux_a: Optional[torch.Tensor] = None, aux_b: Optional[torch.Tensor] = None) -> None:
    n, c, tiles, _ = part.shape
    lib().call("san_norm_finalize", _p(part), n, c, tiles, mode, float(eps), _p(gamma), _p(beta), _p(scale), _p(shift),
               int(scale.shape[1]), coff, _p(aux_a), _p(aux_b), _stream())


def norm_finalize_pool(part: torch.Tensor, eps: float, x: Act, y: Act) -> None:
    """InstanceNorm finalisation of x's records (into x.scale / x.shift) and y = avg_pool2d(lrelu(IN(x))) in one launch
    (san_norm_finalize_pool): the same bits as norm_finalize + avgpool2."""
    n, c, tiles, _ = part.shape
    assert c == x.c == y.c and (x.h, x.w) == (2 * y.h, 2 * y.w) and x.scale is not None
    lib().call("san_norm_finalize_pool", _p(part), n, c, tiles, float(eps), _p(x.scale), _p(x.shift), int(x.scale.shape[1]), x.coff,
               _p(x.buf), x.ctot, x.coff, float(x.slope), _p(y.buf), y.ctot, y.coff, x.h, x.w, _stream())


def replicate_channel(src: Act, dsts, ch: int) -> None:
    """Channel ``ch`` of ``src`` -- raw plane and its (scale, shift) entries -- copied into every Act of ``dsts`` (same shapes), 16
    per launch (san_replicate_channel): the cascades' shared, once-normalised reference channel."""
    n, ctot, hw = src.n, src.ctot, src.h * src.w
    for k0 in range(0, len(dsts), 16):
        grp = dsts[k0:k0 + 16]
        for d in grp:
            assert tuple(d.buf.shape) == tuple(src.buf.shape) and d.scale is not None and tuple(d.scale.shape) == tuple(src.scale.shape)
        arr = (ctypes.c_void_p * (3 * len(grp)))(*([d.buf.data_ptr() for d in grp] + [d.scale.data_ptr() for d in grp] +
                                                  [d.shift.data_ptr() for d in grp]))
        if _lib.KEEP is not None:
            _lib.KEEP.append(arr)               # host memory the recorded call reads again at every replay
        base, step = ctypes.addressof(arr), 8 * len(grp)
        lib().call("san_replicate_channel", _p(src.buf), _p(src.scale), _p(src.shift), base, base + step, base + 2 * step, len(grp),
                   n, ctot, int(ch), hw, _stream())


def plane_stats(x: Act, arena: Arena = GLOBAL_ARENA, tag: str = "") -> torch.Tensor:
    tiles = lib().query("san_plane_stat_tiles", x.h * x.w)
    part = arena.get("pstat" + tag, (x.n, x.c, tiles, 3), x.buf.device)
    lib().call("san_plane_stats", _p(x.buf), x.ctot, x.coff, x.c, x.n, x.h * x.w, _p(part), _stream())
    return part


def bn_eval_affine(gamma, beta, rmean, rvar, eps: float, scale: torch.Tensor, shift: torch.Tensor, coff: int) -> None:
    n, c = scale.shape[0], gamma.shape[0]
    lib().call("san_bn_eval_affine", _p(gamma), _p(beta), _p(rmean), _p(rvar), float(eps), _p(scale), _p(shift),
               int(scale.shape[1]), coff, n, c, _stream())


def avgpool2(x: Act, y: Act) -> None:
    assert x.c == y.c
    lib().call("san_avgpool2_fwd", _p(x.buf), x.ctot, x.coff, _p(x.scale), _p(x.shift), float(x.slope),
               _p(y.buf), y.ctot, y.coff, x.n, x.c, x.h, x.w, _stream())


def upsample2(x: Act, y: Act) -> None:
    assert x.c == y.c
    lib().call("san_upsample2_fwd", _p(x.buf), x.ctot, x.coff, _p(x.scale), _p(x.shift), float(x.slope),
               _p(y.buf), y.ctot, y.coff, x.n, x.c, x.h, x.w, _stream())


def add(a: Act, b: Act, y: Act) -> None:
    assert a.c == b.c == y.c
    lib().call("san_add_fwd", _p(a.buf), a.ctot, a.coff, _p(a.scale), _p(a.shift), float(a.slope),
               _p(b.buf), b.ctot, b.coff, _p(b.scale), _p(b.shift), float(b.slope),
               _p(y.buf), y.ctot, y.coff, a.n, a.c, a.h * a.w, _stream())


def apply(x: Act, y: Act) -> None:
    assert x.c == y.c
    lib().call("san_apply_fwd", _p(x.buf), x.ctot, x.coff, _p(x.scale), _p(x.shift), float(x.slope),
               _p(y.buf), y.ctot, y.coff, x.n, x.c, x.h * x.w, _stream())


def window_copy(x: Act, y: Act, off_y: int = 0, off_x: int = 0, mode: int = 0) -> None:
    """y[.., oy, ox] = T(x)[.., oy - off_y, ox - off_x] between planes of different sizes (san_window_copy_fwd).
    mode 0: zeros outside x (zero pad with off >= 0, crop with off <= 0); 1: reflect one row / column at the bottom /
    right; 2: the adjoint of mode 1."""
    assert x.c == y.c and x.n == y.n
    lib().call("san_window_copy_fwd", _p(x.buf), x.ctot, x.coff, _p(x.scale), _p(x.shift), float(x.slope), x.h, x.w,
               _p(y.buf), y.ctot, y.coff, y.h, y.w, int(off_y), int(off_x), int(mode), x.n, x.c, _stream())


# ---------------------------------------------------------------------------
# warp and losses
# ---------------------------------------------------------------------------
def warp(img: torch.Tensor, offset_nchw: torch.Tensor, padding: str = "zeros",
         want_grid: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Bilinear sample of img [N,C,H,W] at identity + offset (offset NCHW [N,2,H,W])."""
    _chk(img, name="img")
    _chk(offset_nchw, name="offset")
    n, c, h, w = img.shape
    out = torch.empty_like(img)
    grid = torch.empty((n, h, w, 2), device=img.device, dtype=torch.float32) if want_grid else None
    lib().call("san_warp_fwd", _p(img), _p(offset_nchw), _p(out), _p(grid), n, c, h, w,
               0 if padding == "zeros" else 1, _stream())
    return out, grid


def grid_sample(img: torch.Tensor, grid: torch.Tensor, padding: str = "zeros") -> torch.Tensor:
    _chk(img, name="img")
    _chk(grid, name="grid")
    n, c, h, w = img.shape
    ho, wo = grid.shape[1:3]
    out = torch.empty((n, c, ho, wo), device=img.device, dtype=torch.float32)
    lib().call("san_grid_sample_fwd", _p(img), _p(grid), _p(out), n, c, h, w, ho, wo,
               0 if padding == "zeros" else 1, _stream())
    return out


def grid_sample_complex(img: torch.Tensor, grid: torch.Tensor, padding: str = "zeros") -> torch.Tensor:
    """grid_sample on a complex64 image: real and imaginary parts with one grid, one launch."""
    _chk(grid, name="grid")
    n, c, h, w = img.shape
    ho, wo = grid.shape[1:3]
    out = torch.empty((n, c, ho, wo), device=img.device, dtype=torch.complex64)
    lib().call("san_grid_sample_complex_fwd", _p(_creal(img, "img")), _p(grid), _p(_creal(out, "out")), n, c, h, w, ho, wo,
               0 if padding == "zeros" else 1, _stream())
    return out


def augment_grid(affine: torch.Tensor, ctrl: Optional[torch.Tensor], h: int, w: int) -> torch.Tensor:
    """NHWC sampling grid = affine_grid(affine [N,2,3]) + bicubic upsample of ctrl [N,2,cg,cg] (or None)."""
    _chk(affine, name="affine")
    n = affine.shape[0]
    assert tuple(affine.shape) == (n, 2, 3)
    cg = 0
    if ctrl is not None:
        _chk(ctrl, name="ctrl")
        assert ctrl.shape[0] == n and ctrl.shape[1] == 2 and ctrl.shape[2] == ctrl.shape[3]
        cg = int(ctrl.shape[2])
    grid = torch.empty((n, h, w, 2), device=affine.device, dtype=torch.float32)
    lib().call("san_augment_grid", _p(affine), _p(ctrl), _p(grid), n, h, w, cg, _stream())
    return grid


def image_metrics(gt: torch.Tensor, pred: torch.Tensor, bins: int = 64) -> torch.Tensor:
    """[planes, 4] float64: per image sum sq err, sum abs err, sum gt^2, mutual information."""
    _chk(gt, name="gt")
    _chk(pred, name="pred")
    planes = gt.shape[0] * gt.shape[1]
    hw = gt.shape[2] * gt.shape[3]
    out = torch.empty((planes, 4), device=gt.device, dtype=torch.float64)
    lib().call("san_image_metrics", _p(gt), _p(pred), _p(out), planes, hw, int(bins), _stream())
    return out


def _loss_ws(n, h, w, device):
    return GLOBAL_ARENA.get("loss_ws", (lib().query("san_loss_workspace_floats", n, h, w),), device)


def ssim_loss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    _chk(x, name="x")
    _chk(y, name="y")
    n, c, h, w = x.shape
    assert c == 1 and y.shape == x.shape
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    lib().call("san_ssim_loss_fwd", _p(x), _p(y), _p(loss), n, h, w, _p(_loss_ws(n, h, w, x.device)), _stream())
    return loss


def lncc_loss(i: torch.Tensor, j: torch.Tensor, win: int = 9) -> torch.Tensor:
    _chk(i, name="i")
    _chk(j, name="j")
    n, c, h, w = i.shape
    assert c == 1 and j.shape == i.shape
    loss = torch.empty((), device=i.device, dtype=torch.float32)
    lib().call("san_lncc_loss_fwd", _p(i), _p(j), _p(loss), n, h, w, win, _p(_loss_ws(n, h, w, i.device)), _stream())
    return loss


def gradient_loss_nchw(offset_nchw: torch.Tensor) -> torch.Tensor:
    _chk(offset_nchw, name="offset")
    n, two, h, w = offset_nchw.shape
    assert two == 2
    loss = torch.empty((), device=offset_nchw.device, dtype=torch.float32)
    lib().call("san_gradient_loss_fwd", _p(offset_nchw), _p(loss), n, h, w, _p(_loss_ws(n, h, w, offset_nchw.device)),
               _stream())
    return loss


def smooth_pool(x: torch.Tensor, kern: torch.Tensor) -> torch.Tensor:
    """avg_pool2(conv2d(x, kern, padding=k//2)) for a [N,1,H,W] tensor (multi-scale LNCC step)."""
    _chk(x, name="x")
    _chk(kern, name="kern")
    n, c, h, w = x.shape
    k = kern.shape[-1]
    y = torch.empty((n, c, h // 2, w // 2), device=x.device, dtype=torch.float32)
    lib().call("san_smooth_pool_fwd", _p(x), _p(kern), _p(y), n * c, h, w, k, _stream())
    return y


# ---------------------------------------------------------------------------
# backward building blocks
# ---------------------------------------------------------------------------
def packed_weight_dgrad(w: torch.Tensor) -> torch.Tensor:
    """Flipped / channel-swapped packing of a Conv2d weight: conv2d(dy, this) = dL/dx."""
    return PACKS.get(w, 2)


def conv2d_dgrad(dy: Act, weight: torch.Tensor, dx: Act) -> None:
    """dx.buf[:, dx.coff:+cin] = dL/d(conv input) for dy = dL/d(conv output) (materialised)."""
    cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[2]
    assert dy.c == cout and dx.c == cin
    small = (STREAM_SMALL_COUT[0] and ks == 3 and cin in (2, 3) and cout >= 16 and USE_BF16X3[0] and dy.amax is not None and _CONV_NP[0] == 3
             and dy.scale is None and dx.buf.data_ptr() % 16 == 0
             and lib().query("san_conv_stream_eligible", dy.n, dy.h, dy.w, cout, cin, dy.ctot))
    if small or bf16x3_eligible(cout, cin, dy.h, dy.w, ks):       # the data-gradient conv maps cout -> cin channels
        if dy.amax is not None and _CONV_NP[0] == 3 and dy.scale is None:
            wp = PACKS16.get(weight, 2 + 16)             # two fp16 parts; dy scaled by the power of two its maximum asks for
            nbytes = lib().query("san_conv_bf16x3_ws_bytes", dy.n, dy.h, dy.w, cout, cin, 3) if ks == 3 else 0
            ws = GLOBAL_ARENA.scratch("b16_splitk", nbytes, dy.buf.device) if nbytes else None
            gargs = (_p(dy.buf), dy.ctot, dy.coff, cout, _p(wp), _p(dx.buf), dx.ctot, dx.coff, cin, _p(dy.amax), dy.n, dy.h, dy.w,
                     ks, _p(ws), nbytes, _stream())
            fam = "conv3x3_few_cout" if small else ("conv3x3_bf16x3" if ks == 3 else "conv1x1_bf16x3")
            _timed(fam, 2.0 * dy.n * dy.h * dy.w * cout * cin * ks * ks, "FLOP",
                   lambda: lib().call("san_conv_bf16x3_dgrad_amax", *gargs), _conv_abytes(dy.n, dy.h, dy.w, cin, cout, ks), 3)
            return
        wp = PACKS16.get(weight, 2)
        bargs = (_p(dy.buf), dy.ctot, dy.coff, cout, _p(dy.scale), _p(dy.shift), float(dy.slope), _p(wp), _p(None),
                 _p(dx.buf), dx.ctot, dx.coff, cin, _p(None), dy.n, dy.h, dy.w, _stream())
        _timed("conv3x3_bf16x3" if ks == 3 else "conv1x1_bf16x3", 2.0 * dy.n * dy.h * dy.w * cout * cin * ks * ks, "FLOP",
               lambda: _bf16x3_launch(ks, bargs, dy.n, dy.h, dy.w, cout, cin, dy.buf.device, GLOBAL_ARENA),
               _conv_abytes(dy.n, dy.h, dy.w, cin, cout, ks), _products())
        return
    wp = packed_weight_dgrad(weight)
    args = (_p(dy.buf), dy.ctot, dy.coff, cout, _p(dy.scale), _p(dy.shift), float(dy.slope), _p(wp), _p(None),
            _p(dx.buf), dx.ctot, dx.coff, cin, _p(None), _p(None), _p(None), dy.n, dy.h, dy.w, ks, _stream())
    # the data gradient runs on the forward conv kernel: same roofline class
    _timed("conv3x3" if ks == 3 else "conv1x1", 2.0 * dy.n * dy.h * dy.w * cout * cin * ks * ks, "FLOP",
           lambda: lib().call("san_conv2d_fwd", *args), _conv_abytes(dy.n, dy.h, dy.w, cin, cout, ks))


# ---------------------------------------------------------------------------
# Weight gradients on a side stream.  Nothing in the backward chain consumes a weight gradient (only the optimiser
# does), so inside ``with wgrad_overlap():`` every conv2d_wgrad* call is enqueued on a second HIP stream after the
# work that produced its operands, and the main stream goes straight on to the data gradient and the HBM-bound
# activation-backward kernels, which share the CUs with the matrix-bound weight gradient.  Hazards:
#   * dy buffers are arena temporaries the main stream rewrites a layer later -> wgrad_dy_buffer() hands out up to
#     DY_COPIES rotating copies and makes the main stream wait only when all of them still have a reader in flight;
#     Arena.get() applies the same wait to any other buffer with a pending side-stream reader;
#   * allocator-owned operands get record_stream();
#   * leaving the context joins the side stream back into the main one (before the all-reduce / optimiser).
# ---------------------------------------------------------------------------
_WG = {"stream": None, "main": None, "pool": {}, "busy": {}, "rr": {}, "defer": False, "defer_k": 0, "defer_dw": set(),
       "pending": [], "pending_ptrs": set(), "seq": 0, "waited": 0}
# Hand-offs between the two streams cost the GPU: an event record on the main stream is a ~2 us bubble in front of the next
# kernel, a wait ~1-3 us (scratch/event_bubble.py); one pair per weight gradient, ~330 per step, plus the reverse waits of the
# rotating dy copies, were 2 ms of a 46 ms step (measured by leaving them out: gpurun_out/r4b_unsafe.txt).  So weight gradients
# are handed over in BATCHES of up to WGRAD_BATCH launches behind ONE event, and a reverse wait is skipped when the main stream
# already waited for a later batch (the side stream runs in order).  A queued launch keeps its operands alive; an arena buffer
# that a queued launch will read is never handed out again before the queue is flushed.
DY_COPIES = [int(os.environ.get("SAN_DY_COPIES", "8"))]      # rotating copies of a dy buffer that a weight gradient reads (4: the main stream stalled 50-135 us behind every bottleneck-level split-K join; 4 / 6 / 8 / 16: 46.38 / 45.90 / 45.80 / 45.92 ms per step)
WGRAD_BATCH = [int(os.environ.get("SAN_WGRAD_BATCH", "4"))]     # 1: 45.95, 2: 45.54, 4: 45.38, 8: 46.38, 16: 48.5 ms per step (later launches overlap less)
WGRAD_OVERLAP = [os.environ.get("SAN_NO_WGRAD_OVERLAP", "0") != "1"]
# Deferred weight-gradient reductions (san_wgrad_defer): inside wgrad_overlap the side-stream weight gradients queue the
# fixed-order reduction of their partial tiles, and one launch reduces up to 48 layers (each with its own scratch copy).
WGRAD_DEFER = [os.environ.get("SAN_NO_WGRAD_DEFER", "0") != "1"]
_DEFER_BATCH = 48


def wgrad_flush() -> None:
    """Hand over the queued weight gradients and launch the queued reductions (on the side stream)."""
    _flush_pending()
    if _WG["defer"] and _WG["defer_k"]:
        lib().call("san_wgrad_defer_flush", _WG["stream"].cuda_stream)
    _WG["defer_k"] = 0
    _WG["defer_dw"].clear()


def _wgrad_scratch_tag(dw: torch.Tensor, scratch_tag: str):
    """(scratch tag, restore) for one matrix-core weight gradient.  Deferred mode (side stream, no caller tag): a scratch copy
    of its own until the flush; in-line callers (their own tag, main stream) reduce at once."""
    if not _WG["defer"]:
        return scratch_tag, None
    if scratch_tag:
        lib().query("san_wgrad_defer", 0)
        return scratch_tag, (lambda: lib().query("san_wgrad_defer", 1))
    if _WG["defer_k"] >= _DEFER_BATCH or dw.data_ptr() in _WG["defer_dw"]:
        wgrad_flush()
    k = _WG["defer_k"]
    _WG["defer_k"] = k + 1
    _WG["defer_dw"].add(dw.data_ptr())
    return f"@{k}", None


class wgrad_overlap:
    def __enter__(self):
        if WGRAD_OVERLAP[0]:
            dev = torch.cuda.current_device()
            if dev not in _WG["pool"]:
                _WG["pool"][dev] = torch.cuda.Stream(device=dev)
            _WG["stream"] = _WG["pool"][dev]
            _WG["main"] = torch.cuda.current_stream()
            if WGRAD_DEFER[0]:
                lib().query("san_wgrad_defer", 1)
                _WG["defer"], _WG["defer_k"] = True, 0
                _WG["defer_dw"].clear()
        return self

    def __exit__(self, *exc):
        side = _WG["stream"]
        if side is not None:
            if exc and exc[0] is not None:      # the step failed: its queued weight gradients are dropped, not launched
                _WG["pending"] = []
                _WG["pending_ptrs"] = set()
            _flush_pending()
            if _WG["defer"]:
                wgrad_flush()
                lib().query("san_wgrad_defer", 0)
                _WG["defer"] = False
            _lib.rec(torch.cuda.current_stream().wait_stream, side)
        _WG["stream"] = None
        _WG["main"] = None
        _WG["busy"].clear()
        _WG["rr"].clear()
        _WG["seq"] = _WG["waited"] = 0
        return False


class aux_region:
    """``with aux_region(stream, arena):`` -- an independent branch of the step (the sensitivity network beside the alignment
    network: model.py) issued on ``stream`` with its own arena.  Launches inside take ``stream`` (it is torch's current stream
    for the block); arena requests go to ``arena``, so nothing is shared with the main branch's temporaries; weight gradients
    run in line on ``stream`` with their reductions at once (the side stream's hand-offs are ordered against the MAIN stream,
    and the library's deferred-reduction queue belongs to it).  The caller orders the region against its producers and consumers
    (``stream.wait_stream(main)`` before, ``main.wait_stream(stream)`` where the results are read)."""

    def __init__(self, stream, arena: Arena):
        self.stream, self.arena = stream, arena

    def __enter__(self):
        self.saved = (_WG["stream"], _WG["defer"])
        if _WG["defer"]:
            lib().query("san_wgrad_defer", 0)
        _WG["stream"], _WG["defer"] = None, False
        self.prev_aux = _lib.AUX[0]
        _lib.AUX[0] = self.stream
        self.cm = torch.cuda.stream(self.stream)
        self.cm.__enter__()
        self.ua = use_arena(self.arena)
        self.ua.__enter__()
        return self

    def __exit__(self, *exc):
        self.ua.__exit__(*exc)
        self.cm.__exit__(*exc)
        _lib.AUX[0] = self.prev_aux
        _WG["stream"], _WG["defer"] = self.saved
        if _WG["defer"]:
            lib().query("san_wgrad_defer", 1)
        return False


def ensure_packs(device) -> None:
    """Re-pack the registered weight images NOW (on the current stream) if an optimiser step made them stale -- instead of at the
    first convolution that asks, which inside an aux_region would put the batch on the wrong stream."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:  # ('cuda' == the current device: the jobs carry their weights' indexed device)
        device = torch.device("cuda", torch.cuda.current_device())
    for reg in (PACKS, PACKS16):
        if reg.jobs and reg.epoch != WEIGHT_EPOCH[0]:
            reg._run(device)


class backward_scope:
    """Everything one backward pass needs around it, entered by whoever starts it (CSModel.update, VarNet.backward,
    SpatialTransformer.backward, the autograd Functions): the gradient-maximum records of the previous pass are zeroed
    (AmaxPool.reset) and the weight gradients go to the side stream (wgrad_overlap), joined on exit.  Nested scopes are
    no-ops, so a module-level backward inside CSModel.update shares the step's scope."""
    _depth = [0]

    def __init__(self, device):
        self.device = device
        self.ov = None

    def __enter__(self):
        self._depth[0] += 1
        if self._depth[0] == 1:
            AMAX.reset(self.device)
            self.ov = wgrad_overlap()
            self.ov.__enter__()
        return self

    def __exit__(self, *exc):
        self._depth[0] -= 1
        if self.ov is not None:
            self.ov.__exit__(*exc)
            self.ov = None
        return False


def _wait_if_busy(t: torch.Tensor) -> None:
    ptr = t.data_ptr()
    if ptr in _WG["pending_ptrs"]:              # a queued weight gradient will read it: hand the queue over first
        _flush_pending()
    ent = _WG["busy"].pop(ptr, None)
    if ent is not None and ent[0] > _WG["waited"]:      # (a wait for a later batch covers every earlier one: the side stream is in order)
        _WG["waited"] = ent[0]
        _lib.rec(torch.cuda.current_stream().wait_event, ent[1])


def wgrad_dy_buffer(name: str, shape, device, arena: Arena = GLOBAL_ARENA) -> torch.Tensor:
    """An arena temporary that a weight gradient will read: the plain arena buffer outside wgrad_overlap, otherwise
    one of up to DY_COPIES rotating copies without a side-stream reader in flight (else the oldest, after waiting)."""
    if _WG["stream"] is None:
        return arena.get(name, shape, device)
    if torch.cuda.is_current_stream_capturing() or _lib.REC is not None:
        # hipGraph capture / step recording: no event queries (the choice must not depend on timing); take the copies
        # round-robin and wait (a stream dependency, not a host wait) for the reader that used this copy four requests ago
        k = _WG["rr"].get(name, 0)
        _WG["rr"][name] = (k + 1) % DY_COPIES[0]
        t = arena.get(name if k == 0 else f"{name}#{k}", shape, device, _no_wait=True)
        _wait_if_busy(t)
        return t
    first = None
    for k in range(DY_COPIES[0]):
        t = arena.get(name if k == 0 else f"{name}#{k}", shape, device, _no_wait=True)
        if t.data_ptr() in _WG["pending_ptrs"]:
            if first is None:
                first = t
            continue
        ent = _WG["busy"].get(t.data_ptr())
        if ent is None or ent[0] <= _WG["waited"] or ent[1].query():
            _WG["busy"].pop(t.data_ptr(), None)
            return t
        if first is None:
            first = t
    _wait_if_busy(first)
    return first


class _EventRing:
    """Reusable events for the main <-> side stream hand-offs (creating a torch.cuda.Event per hand-off, two per weight
    gradient, ~650 per step, was ~1 ms of host time).  A stream wait captures the event's LAST record at the time of the
    call, so an event may be recorded again once its waits have been issued; the ring is far longer than a step."""

    def __init__(self, n: int = 4096):
        self.n, self.i, self.ev = n, 0, {}

    def next(self) -> "torch.cuda.Event":
        dev = torch.cuda.current_device()
        ring = self.ev.get(dev)
        if ring is None:
            ring = self.ev[dev] = [torch.cuda.Event() for _ in range(self.n)]
        self.i = (self.i + 1) % self.n
        return ring[self.i]


_EVENTS = _EventRing()          # main-stream hand-off marks ('e0': waited on by the side stream right away)
_BUSY_EVENTS = _EventRing()     # side-stream completion marks: _WG['busy'] may hold one across many later hand-offs, so they come from
                                # their own ring and can never be re-recorded as a main-stream mark


def _on_side_stream(dy: Act, x: Act, fn) -> None:
    """``fn(x, dy)`` launches one weight gradient.  Outside wgrad_overlap it runs at once; inside it is queued with a SNAPSHOT of
    what it will read at launch time: private copies of the two Act records (a caller may re-point .buf / .coff / .amax of the
    objects it passed before the queue is handed over) and the convolution precision in force now."""
    side = _WG["stream"]
    if side is None:
        fn(x, dy)
        return
    dy, x = dataclasses.replace(dy), dataclasses.replace(x)
    mode = _CONV_NP[0], _FP8_FWD[0]

    def launch(fn=fn, x=x, dy=dy, mode=mode):
        saved = _CONV_NP[0], _FP8_FWD[0]
        _CONV_NP[0], _FP8_FWD[0] = mode
        try:
            fn(x, dy)
        finally:
            _CONV_NP[0], _FP8_FWD[0] = saved

    _WG["pending"].append((dy, x, launch))
    _WG["pending_ptrs"].add(dy.buf.data_ptr())
    _WG["pending_ptrs"].add(x.buf.data_ptr())
    if len(_WG["pending"]) >= WGRAD_BATCH[0]:
        _flush_pending()


def _flush_pending() -> None:
    """Hand the queued weight gradients to the side stream: one event on the main stream (their operands' producers are queued
    on main up to here), the launches in the order they were asked for, one completion mark for all their operands."""
    pend = _WG["pending"]
    side = _WG["stream"]
    if not pend or side is None:
        return
    _WG["pending"] = []
    _WG["pending_ptrs"] = set()
    main = _WG["main"]
    if torch.cuda.is_current_stream_capturing():
        side.wait_stream(main)                  # (capture: fresh events, the graph keeps them as edges)
    else:
        e0 = _lib.light(torch.cuda.Event()) if _lib.REC is not None else _EVENTS.next()      # (a recorded step owns its events)
        _lib.rec(e0.record, main)
        _lib.rec(side.wait_event, e0)
    _STREAM_OVERRIDE[0] = side                  # (launches take the side stream's handle; torch's current stream stays put)
    try:
        for _, _, fn in pend:
            fn()
    finally:
        _STREAM_OVERRIDE[0] = None
    ev = _lib.light(torch.cuda.Event()) if (torch.cuda.is_current_stream_capturing() or _lib.REC is not None) else _BUSY_EVENTS.next()
    _lib.rec(ev.record, side)
    _WG["seq"] += 1
    ent = (_WG["seq"], ev)
    for dy, x, _ in pend:
        dptr = dy.buf.data_ptr()
        _WG["busy"][dptr] = ent
        if dptr not in _ARENA_PTRS:             # allocator-owned operands: keep their memory until the side stream is done
            dy.buf.record_stream(side)
        if x.buf.data_ptr() not in _ARENA_PTRS:
            x.buf.record_stream(side)


def conv2d_wgrad(x: Act, dy: Act, dw: torch.Tensor, accumulate: bool = False, arena: Arena = GLOBAL_ARENA) -> None:
    """dw [cout, cin, ks, ks] (+)= correlation of dy with the lazily activated forward input x (on the side stream
    inside ``wgrad_overlap``)."""
    _on_side_stream(dy, x, lambda x, dy: _conv2d_wgrad(x, dy, dw, accumulate, arena))


def _conv2d_wgrad(x: Act, dy: Act, dw: torch.Tensor, accumulate: bool = False, arena: Arena = GLOBAL_ARENA,
                  scratch_tag: str = "") -> None:
    """dw [cout, cin, ks, ks] (+)= correlation of dy with the lazily activated forward input x.  ``scratch_tag``: callers
    that run IN LINE on the main stream inside wgrad_overlap pass their own tag, so that their partial-sum scratch is
    not the buffer a side-stream weight gradient may still be using."""
    cout, cin, ks = dw.shape[0], dw.shape[1], dw.shape[2]
    assert x.c == cin and dy.c == cout and x.buf.shape[2:] == dy.buf.shape[2:]
    flops = 2.0 * x.n * x.h * x.w * cout * cin * ks * ks
    if USE_BF16X3[0] and lib().query("san_conv_wgrad_bf16x3_eligible", x.n, x.h, x.w, cin, cout, ks):
        _conv2d_wgrad_bf16x3(x, dy, dw, accumulate, arena, scratch_tag)
        return
    if ks == 1 and wgrad1x1_bf16x3_ok(x, dy):
        _conv2d_wgrad1x1_bf16x3(x, dy, dw, accumulate, arena, scratch_tag=scratch_tag)
        return
    P = lib().query("san_conv_wgrad_partitions", x.n, x.h, x.w, cin, cout, ks)
    partial = arena.get("wgrad_partial" + scratch_tag, (P * cout * cin * ks * ks,), x.buf.device)
    args = (_p(x.buf), x.ctot, x.coff, cin, _p(x.scale), _p(x.shift), float(x.slope), _p(dy.buf), dy.ctot, dy.coff, cout,
            _p(_chk(dw, name="dw")), int(accumulate), _p(partial), x.n, x.h, x.w, ks, _stream())
    _timed("wgrad3x3" if ks == 3 else "wgrad1x1", flops, "FLOP", lambda: lib().call("san_conv2d_wgrad", *args),
           _conv_abytes(x.n, x.h, x.w, cin, cout, ks))


def conv2d_wgrad_bf16x3(x: Act, dy: Act, dw: torch.Tensor, accumulate: bool = False, arena: Arena = GLOBAL_ARENA) -> None:
    _on_side_stream(dy, x, lambda x, dy: _conv2d_wgrad_bf16x3(x, dy, dw, accumulate, arena))


def _conv2d_wgrad_bf16x3(x: Act, dy: Act, dw: torch.Tensor, accumulate: bool = False, arena: Arena = GLOBAL_ARENA,
                         scratch_tag: str = "") -> None:
    """The 3x3 weight gradient on the bf16 matrix cores (three-way split operands, fp32-level accuracy;
    csrc/san_wgrad_bf16.hip).  conv2d_wgrad dispatches here where san_conv_wgrad_bf16x3_eligible says so."""
    cout, cin, ks = dw.shape[0], dw.shape[1], dw.shape[2]
    assert ks == 3 and x.c == cin and dy.c == cout and x.buf.shape[2:] == dy.buf.shape[2:]
    if not lib().query("san_conv_wgrad_bf16x3_supported", x.n, x.h, x.w, cin, cout, ks):
        raise RuntimeError("layer too large for the bf16x3 weight gradient")
    nbytes = lib().query("san_conv_wgrad_bf16x3_scratch_bytes", x.n, x.h, x.w, cin, cout)
    scratch_tag, restore = _wgrad_scratch_tag(dw, scratch_tag)
    scratch = arena.scratch("wgrad_bf16x3" + scratch_tag, nbytes, x.buf.device)
    f16 = dy.amax is not None and _CONV_NP[0] == 3
    args = (_p(x.buf), x.ctot, x.coff, cin, _p(x.scale), _p(x.shift), float(x.slope), _p(dy.buf), dy.ctot, dy.coff,
            cout, _p(_chk(dw, name="dw")), int(accumulate), _p(scratch)) + ((_p(dy.amax),) if f16 else ()) + (x.n, x.h, x.w, _stream())
    fn = "san_conv2d_wgrad_bf16x3_amax" if f16 else "san_conv2d_wgrad_bf16x3"
    try:
        _timed("wgrad3x3_bf16x3", 2.0 * x.n * x.h * x.w * cout * cin * 9, "FLOP", lambda: lib().call(fn, *args),
               _conv_abytes(x.n, x.h, x.w, cin, cout, 3), _products(f16))
    finally:
        if restore is not None:
            restore()


def wgrad1x1_bf16x3_ok(x: Act, dy: Act) -> bool:
    return bool(USE_BF16X3[0] and lib().query("san_conv1x1_wgrad_bf16x3_eligible", x.n, x.h, x.w, x.c, dy.c)
                and (x.buf.data_ptr() | dy.buf.data_ptr()) % 16 == 0)


def conv2d_wgrad1x1_bf16x3(x: Act, dy: Act, dw: torch.Tensor, accumulate: bool = False, arena: Arena = GLOBAL_ARENA,
                           transposed: bool = False) -> None:
    _on_side_stream(dy, x, lambda x, dy: _conv2d_wgrad1x1_bf16x3(x, dy, dw, accumulate, arena, transposed))


def _conv2d_wgrad1x1_bf16x3(x: Act, dy: Act, dw: torch.Tensor, accumulate: bool = False, arena: Arena = GLOBAL_ARENA,
                            transposed: bool = False, scratch_tag: str = "") -> None:
    """The 1x1 weight gradient on the bf16 matrix cores (csrc/san_wgrad_bf16.hip, wgrad1x1_bf16x3_kernel).
    transposed: dw is laid out [cin, cout(, ...)] -- a ConvTranspose2d weight [Cin, Cout, 2, 2] seen as [Cin, 4 Cout]."""
    cin, cout = x.c, dy.c
    assert dw.numel() == cin * cout and x.buf.shape[2:] == dy.buf.shape[2:]
    assert transposed or (dw.shape[0], dw.shape[1]) == (cout, cin)
    nbytes = lib().query("san_conv1x1_wgrad_bf16x3_scratch_bytes", x.n, x.h, x.w, cin, cout)
    scratch_tag, restore = _wgrad_scratch_tag(dw, scratch_tag)
    scratch = arena.scratch("wgrad_bf16x3" + scratch_tag, nbytes, x.buf.device)
    f16 = dy.amax is not None and _CONV_NP[0] == 3
    args = (_p(x.buf), x.ctot, x.coff, cin, _p(x.scale), _p(x.shift), float(x.slope), _p(dy.buf), dy.ctot, dy.coff,
            cout, _p(_chk(dw, name="dw")), int(accumulate), int(transposed), _p(scratch)) + ((_p(dy.amax),) if f16 else ()) + (
            x.n, x.h, x.w, _stream())
    fn = "san_conv1x1_wgrad_bf16x3_amax" if f16 else "san_conv1x1_wgrad_bf16x3"
    try:
        _timed("wgrad1x1_bf16x3", 2.0 * x.n * x.h * x.w * cout * cin, "FLOP", lambda: lib().call(fn, *args))
    finally:
        if restore is not None:
            restore()


def _act_bwd_bytes(y: Act, instance_norm: bool, extra: float = 0.0) -> float:
    """HBM bytes one norm + LeakyReLU backward call really moves: planes up to 160^2 are read once (g, y) and written once (one
    workgroup holds the plane); larger planes take the two reductions and the apply pass separately (g, y twice + dy); without
    a norm there are no reductions.  ``extra`` = further planes (the half-resolution second source, an accumulated destination)."""
    planes = 3.0 if (not instance_norm or y.h * y.w <= 160 * 160) else 5.0
    return 4.0 * y.n * y.c * y.h * y.w * (planes + extra)


def act_bwd_up_ok(y: Act) -> bool:
    """True where act_bwd can take a half-resolution second gradient source (even height, width % 4 == 0)."""
    return (y.h & 1) == 0 and (y.w & 3) == 0 and y.h * y.w // 4 < (1 << 22)


def _act_bwd_in(g: Act, y: Act, dy: Act, arena: Arena, g2: Optional[Act] = None, g2_scale: float = 0.25, flags: int = 0) -> bool:
    """The one-pass cluster form of the InstanceNorm backward (san_act_bwd_in, round 6) where the library wants it for this
    shape and the tensors are 16-byte aligned; False: the caller runs the multi-launch form.  ``dy.amax`` is already set."""
    hw = y.h * y.w
    words = lib().query("san_act_bwd_in_sync_words", y.n, y.c, hw)
    if not words:
        return False
    if (g.buf.data_ptr() | y.buf.data_ptr() | (0 if (flags & 1) else dy.buf.data_ptr())) % 16 or (g2 is not None and g2.buf.data_ptr() % 8):
        return False
    # zero once, at creation: the kernel leaves its counters zero (a recorded step replays the launch with the same buffer)
    # (one buffer per SHAPE, not per size: a record's layout -- members' sums, then the two counters -- depends on the cluster
    # size, and only the counters return to zero; two shapes with equal word counts must never meet in one buffer)
    sync = arena.get(f"bwd_sync.{y.n}.{y.c}.{hw}", (words,), y.buf.device, dtype=torch.int32, zero=True)
    args = (_p(g.buf), g.ctot, g.coff, _p(None if g2 is None else g2.buf), 0 if g2 is None else g2.ctot, 0 if g2 is None else g2.coff,
            float(g2_scale), _p(y.buf), y.ctot, y.coff, _p(y.scale), _p(y.shift), float(y.slope), _p(dy.buf), dy.ctot, dy.coff,
            _p(dy.amax), y.n, y.c, hw, y.w, int(flags), _p(sync), _stream())
    # one pass: g and y read once, dy written once (+ the quarter-size second source / the accumulated destination)
    extra = (0.25 if g2 is not None else 0.0) + (1.0 if (flags & 2) else 0.0)
    _timed("act_bwd", 4.0 * y.n * y.c * hw * (3.0 + extra), "B", lambda: lib().call("san_act_bwd_in", *args))
    return True


def act_bwd_ex(g: Act, y: Act, dy: Act, instance_norm: bool, arena: Arena = GLOBAL_ARENA, unshuffle: bool = False,
               accumulate: bool = False) -> None:
    """act_bwd with a destination mode (san_act_bwd_ex_amax).  unshuffle: dy is a [n, 4c, h/2, w/2] view and receives the
    gradient pixel-unshuffled (channel 4 ch + 2 (row & 1) + (col & 1)): the transposed convolution's backward without the
    separate un-shuffle pass.  accumulate: dy += ."""
    assert g.c == y.c and (dy.c == 4 * y.c if unshuffle else dy.c == y.c)
    hw = y.h * y.w
    part = None
    dy.amax = AMAX.next(y.buf.device)
    if instance_norm and _act_bwd_in(g, y, dy, arena, flags=(1 if unshuffle else 0) | (2 if accumulate else 0)):
        return
    if instance_norm:
        part = arena.get("bwd_part", (y.n, y.c, lib().query("san_bwd_stat_tiles", hw), 2), y.buf.device)
    eargs = (_p(g.buf), g.ctot, g.coff, _p(y.buf), y.ctot, y.coff, _p(y.scale), _p(y.shift),
             float(y.slope), 1 if instance_norm else 0, _p(part), _p(dy.buf), dy.ctot, dy.coff, _p(dy.amax), y.n, y.c, hw, y.w,
             (1 if unshuffle else 0) | (2 if accumulate else 0), _stream())
    _timed("act_bwd", _act_bwd_bytes(y, instance_norm, 1.0 if accumulate else 0.0), "B", lambda: lib().call("san_act_bwd_ex_amax", *eargs))


def act_bwd_unshuffle_ok(y: Act) -> bool:
    return (y.h & 1) == 0 and (y.w & 3) == 0 and y.h * y.w // 4 < (1 << 22) and y.buf.data_ptr() % 16 == 0


def act_bwd(g: Act, y: Act, dy: Act, instance_norm: bool, arena: Arena = GLOBAL_ARENA, g2: Optional[Act] = None,
            g2_scale: float = 0.25) -> None:
    """Gradient through y's lazy (scale, shift, LeakyReLU) read: g = dL/d(activation) -> dy = dL/dy_raw.
    g2 (a materialised [n, c, h/2, w/2] view): the incoming gradient is g(p) + g2_scale * g2(p // 2) -- the encoder levels'
    skip gradient + average-pool adjoint, summed while it is read (act_bwd_up_ok(y) must hold)."""
    assert g.c == y.c == dy.c
    hw = y.h * y.w
    part = None
    dy.amax = AMAX.next(y.buf.device)
    if g2 is not None:
        assert g2.c == y.c and (2 * g2.h, 2 * g2.w) == (y.h, y.w) and act_bwd_up_ok(y)
    if instance_norm and _act_bwd_in(g, y, dy, arena, g2, g2_scale):
        return
    if instance_norm:
        tiles = lib().query("san_bwd_stat_tiles", hw)
        part = arena.get("bwd_part", (y.n, y.c, tiles, 2), y.buf.device)
    if g2 is not None:
        uargs = (_p(g.buf), g.ctot, g.coff, _p(g2.buf), g2.ctot, g2.coff, float(g2_scale), _p(y.buf),
                 y.ctot, y.coff, _p(y.scale), _p(y.shift), float(y.slope), 1 if instance_norm else 0, _p(part), _p(dy.buf),
                 dy.ctot, dy.coff, _p(dy.amax), y.n, y.c, hw, y.w, _stream())
        _timed("act_bwd", _act_bwd_bytes(y, instance_norm, 0.25 * (1.0 if y.h * y.w <= 160 * 160 else 2.0)), "B",
               lambda: lib().call("san_act_bwd_up_amax", *uargs))
        return
    if dy.amax is None:
        aargs = (_p(g.buf), g.ctot, g.coff, _p(y.buf), y.ctot, y.coff, _p(y.scale), _p(y.shift),
                 float(y.slope), 1 if instance_norm else 0, _p(part), _p(dy.buf), dy.ctot, dy.coff, y.n, y.c, hw, _stream())
        _timed("act_bwd", _act_bwd_bytes(y, instance_norm), "B", lambda: lib().call("san_act_bwd", *aargs))
    else:
        aargs = (_p(g.buf), g.ctot, g.coff, _p(y.buf), y.ctot, y.coff, _p(y.scale), _p(y.shift),
                 float(y.slope), 1 if instance_norm else 0, _p(part), _p(dy.buf), dy.ctot, dy.coff, _p(dy.amax),
                 y.n, y.c, hw, _stream())
        _timed("act_bwd", _act_bwd_bytes(y, instance_norm), "B", lambda: lib().call("san_act_bwd_amax", *aargs))


def unshuffle2(x: Act, y: Act) -> None:
    """y[n, 4c+2dy+dx, i, j] = x[n, c, 2i+dy, 2j+dx] (x is read raw: pass a materialised gradient)."""
    assert y.c == 4 * x.c and x.h == 2 * y.h and x.w == 2 * y.w
    lib().call("san_unshuffle2_fwd", _p(x.buf), x.ctot, x.coff, _p(y.buf), y.ctot, y.coff, x.n, x.c, y.h, y.w, _stream())
    y.amax = x.amax                                     # a permutation: the same largest magnitude


def plane_dot_part(g: Act, y: Act, tag: str = "", arena: Arena = GLOBAL_ARENA) -> torch.Tensor:
    """The chunk sums [n, c, tiles, 2] of (u, u*yh) per plane, with yh = y's lazy affine value and u = g * lrelu'(yh):
    the two plane reductions every normalisation backward needs; consumed by the one-launch finalisations."""
    assert g.c == y.c
    hw = y.h * y.w
    tiles = lib().query("san_bwd_stat_tiles", hw)
    part = arena.get("pdot" + tag, (y.n, y.c, tiles, 2), y.buf.device)
    lib().call("san_plane_dot_stats", _p(g.buf), g.ctot, g.coff, _p(y.buf), y.ctot, y.coff, _p(y.scale), _p(y.shift),
               float(y.slope), _p(part), y.n, y.c, hw, _stream())
    return part


def bn_bwd_coef(g: Act, y: Act, gamma: torch.Tensor, beta: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor,
                arena: Arena = GLOBAL_ARENA) -> torch.Tensor:
    """BatchNorm2d training backward (unet.py:125) for the activation y = lrelu(bn(conv)) read lazily: accumulates
    dgamma / dbeta and returns coef [n, c, 4] for act_bwd_coef.  Two launches, no host arithmetic."""
    part = plane_dot_part(g, y, "bn", arena)
    coef = arena.get("bn_coef", (y.n, y.c, 4), y.buf.device)
    lib().call("san_bn_bwd_finalize", _p(part), _p(_chk(gamma, name="gamma")), _p(_chk(beta, name="beta")),
               _p(_chk(dgamma, name="dgamma")), _p(_chk(dbeta, name="dbeta")), _p(coef), y.n, y.c, int(part.shape[2]),
               float(y.n * y.h * y.w), _stream())
    return coef


def bn_act_bwd(g: Act, y: Act, gamma: torch.Tensor, beta: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor, dy: Act,
               arena: Arena = GLOBAL_ARENA) -> None:
    """Training-mode BatchNorm2d + LeakyReLU backward (unet.py:125): dy = dL/dy_raw, dgamma / dbeta accumulated.  ONE launch on
    workgroup clusters (san_bn_act_bwd, round 6) where the library covers the shape, else plane sums + finalisation + apply."""
    assert g.c == y.c == dy.c
    hw = y.h * y.w
    words = lib().query("san_bn_act_bwd_sync_words", y.n, y.c, hw)
    if words and (g.buf.data_ptr() | y.buf.data_ptr() | dy.buf.data_ptr()) % 16 == 0:
        dy.amax = AMAX.next(y.buf.device)
        sync = arena.get(f"bn_bwd_sync.{y.n}.{y.c}.{hw}", (words,), y.buf.device, dtype=torch.int32, zero=True)     # (per shape: see _act_bwd_in)
        lib().call("san_bn_act_bwd", _p(g.buf), g.ctot, g.coff, _p(y.buf), y.ctot, y.coff, _p(y.scale), _p(y.shift), float(y.slope),
                   _p(_chk(gamma, name="gamma")), _p(_chk(beta, name="beta")), _p(_chk(dgamma, name="dgamma")),
                   _p(_chk(dbeta, name="dbeta")), _p(dy.buf), dy.ctot, dy.coff, _p(dy.amax), y.n, y.c, hw, _p(sync), _stream())
        return
    coef = bn_bwd_coef(g, y, gamma, beta, dgamma, dbeta, arena)
    act_bwd_coef(g, y, coef, dy)


def norm_finalize_bn(part: torch.Tensor, eps: float, scale: torch.Tensor, shift: torch.Tensor, coff: int, bn, bmean: torch.Tensor,
                     bvar: torch.Tensor, momentum: float, var_factor: float) -> None:
    """Training-mode BatchNorm2d after a convolution that emitted statistics: the lazy affine, the batch mean / unbiased variance
    (the backward's) and the running statistics + num_batches_tracked, one launch (unet.py:125)."""
    n, c, tiles, _ = part.shape
    lib().call("san_norm_finalize_bn", _p(part), n, c, tiles, float(eps), _p(bn.weight), _p(bn.bias), _p(scale), _p(shift),
               int(scale.shape[1]), coff, _p(bmean), _p(bvar), _p(_chk(bn.running_mean, name="running_mean")),
               _p(_chk(bn.running_var, name="running_var")), _p(_chk(bn.num_batches_tracked, torch.int64, "num_batches_tracked")),
               float(momentum), float(var_factor), _stream())


def bn_update_running(bn, bmean: torch.Tensor, bvar: torch.Tensor, momentum: float, var_factor: float) -> None:
    """BatchNorm2d running statistics and num_batches_tracked in one launch (unet.py:125)."""
    c = bn.running_mean.shape[0]
    lib().call("san_bn_update_running", _p(_chk(bn.running_mean, name="running_mean")), _p(_chk(bn.running_var, name="running_var")),
               _p(_chk(bn.num_batches_tracked, torch.int64, "num_batches_tracked")), _p(_chk(bmean, name="bmean")),
               _p(_chk(bvar, name="bvar")), c, float(momentum), float(var_factor), _stream())


def bias_grad_acc(part: torch.Tensor, db: torch.Tensor) -> None:
    """db[c] += sum of a plane_stats result (count * mean over samples and chunks)."""
    n, c, tiles, _ = part.shape
    lib().call("san_bias_grad_from_stats", _p(part), _p(_chk(db, name="db")), n, c, tiles, _stream())


def normunet_bwd_coefs(part_b: torch.Tensor, part_a: torch.Tensor, xin: Act, std: torch.Tensor, nel: int, g_ctot: int,
                       arena: Arena = GLOBAL_ARENA):
    """The two lazy affines of NormUnet's input gradient (see san_normunet_bwd_coefs): (a_sc, a_sh, m_sc, m_sh)."""
    b, dev = xin.n, xin.buf.device
    a_sc, a_sh = arena.get("nub.a_sc", (b, g_ctot), dev), arena.get("nub.a_sh", (b, g_ctot), dev)
    m_sc, m_sh = arena.get("nub.m_sc", (b, xin.ctot), dev), arena.get("nub.m_sh", (b, xin.ctot), dev)
    assert xin.coff == 0
    lib().call("san_normunet_bwd_coefs", _p(part_b), _p(part_a), int(part_b.shape[2]), _p(xin.scale), _p(xin.shift), xin.ctot,
               _p(_chk(std, name="std")), float(nel), _p(a_sc), _p(a_sh), g_ctot, _p(m_sc), _p(m_sh), b, _stream())
    return a_sc, a_sh, m_sc, m_sh


def normunet_bwd_head(g_out: torch.Tensor, out_planar: torch.Tensor, isd: torch.Tensor, nshift: torch.Tensor, std: torch.Tensor,
                      g_u: torch.Tensor, arena: Arena = GLOBAL_ARENA, tag: str = "nu.b") -> torch.Tensor:
    """NormUnet backward, first launch (san_normunet_bwd_head): g_u = g_out * std and the chunk sums part_b [b, 2, tiles, 2] of
    (g_out, g_out * U) that plane_dot_part would give -- one pass instead of plane_dot_part + apply (+ plane_stats + the bias
    gradient kernel: the tail launch takes the last convolution's bias gradient from part_b)."""
    b, two, h, w = g_out.shape
    assert two == 2 and out_planar.shape == g_out.shape and g_u.shape == g_out.shape
    hw = h * w
    part = arena.get("pdot" + tag, (b, 2, lib().query("san_bwd_stat_tiles", hw), 2), g_out.device)
    lib().call("san_normunet_bwd_head", _p(_chk(g_out, name="g_out")), _p(_chk(out_planar, name="out")), _p(_chk(isd, name="isd")),
               _p(_chk(nshift, name="nshift")), _p(_chk(std, name="std")), _p(_chk(g_u, name="g_u")), _p(part), b, hw, _stream())
    return part


def normunet_bwd_tail(part_b: torch.Tensor, part_x: torch.Tensor, xin: Act, std: torch.Tensor, nel: int, g_xh: torch.Tensor,
                      gd: torch.Tensor, sens: torch.Tensor, gS: Optional[torch.Tensor], r_planar: Optional[torch.Tensor],
                      t1: Optional[torch.Tensor], xs: Optional[torch.Tensor], sign1: float, g_ref: Optional[torch.Tensor],
                      ref_accumulate: bool, db: Optional[torch.Tensor], dcw_part: Optional[torch.Tensor], dcw_scale: float,
                      dcw: Optional[torch.Tensor]) -> None:
    """NormUnet backward, last launch (san_normunet_bwd_tail): the affines of dL/dm, gd += g_m * S (g_m never stored), the
    sensitivity-map accumulation, the reference channel's gradient, the last convolution's bias gradient and dc_weight's."""
    n, c, h, w = gd.shape
    xc = int(part_x.shape[1])
    assert xin.coff == 0 and tuple(part_b.shape[:2]) == (n, 2) and part_x.shape[0] == n and part_x.shape[2] == part_b.shape[2]
    assert g_xh.shape[0] == n and tuple(g_xh.shape[2:]) == (h, w) and tuple(xin.buf.shape[2:]) == (h, w)
    if gS is not None:
        assert r_planar is not None and t1 is not None and xs is not None
    lib().call("san_normunet_bwd_tail", _p(part_b), _p(part_x), xc, _p(xin.buf), xin.ctot, _p(xin.scale), _p(xin.shift),
               _p(_chk(std, name="std")), float(nel), _p(_chk(g_xh, name="g_xh")), int(g_xh.shape[1]), _p(_creal(gd, "gd")),
               _p(_creal(sens, "sens")), _p(None if gS is None else _creal(gS, "gS")),
               _p(None if gS is None else _chk(r_planar, name="r")), _p(None if gS is None else _creal(t1, "t1")),
               _p(None if gS is None else _creal(xs, "x")), float(sign1), _p(None if g_ref is None else _chk(g_ref, name="g_ref")),
               int(ref_accumulate), _p(None if db is None else _chk(db, name="db")), _p(dcw_part),
               0 if dcw_part is None else int(dcw_part.numel()), float(dcw_scale), _p(dcw), n, c, h * w, _stream())


def dc_weight_grad(G: torch.Tensor, k: torch.Tensor, k0: torch.Tensor, mask_f: torch.Tensor) -> torch.Tensor:
    """dL/d(dc_weight) of k' = k - w*M*(k-k0) - R given G = dL/dk' (0-d tensor)."""
    n, c, h, w = k.shape
    part = torch.empty(256, device=k.device, dtype=torch.float32)
    lib().call("san_dc_weight_grad", _p(_creal(G, "G")), _p(_creal(k, "k")), _p(_creal(k0, "k0")), _p(mask_f), _p(part),
               n * c, h, w, _stream())
    return part.double().sum().float()


def sens_grad_acc(gS: torch.Tensor, r_planar: torch.Tensor, t1: torch.Tensor, x: torch.Tensor, gm_planar: torch.Tensor,
                  sign1: float) -> None:
    n, c, h, w = gS.shape
    lib().call("san_sens_grad_acc", _p(_creal(gS, "gS")), _p(_chk(r_planar, name="r")), _p(_creal(t1, "t1")),
               _p(_creal(x, "x")), _p(_chk(gm_planar, name="gm")), float(sign1), n, c, h * w, _stream())


def sens_normalize_bwd(est_planar: torch.Tensor, gS: torch.Tensor) -> torch.Tensor:
    n, c, h, w = gS.shape
    out = torch.empty_like(est_planar)
    lib().call("san_sens_normalize_bwd", _p(_chk(est_planar, name="est")), _p(_creal(gS, "gS")), _p(out), n, c, h * w,
               _stream())
    return out


def rss_bwd(x: torch.Tensor, y: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """gradient of y = rss(x) wrt x (complex or real x)."""
    n, c = x.shape[:2]
    hw = x.shape[2] * x.shape[3]
    gx = torch.empty_like(x)
    if torch.is_complex(x):
        lib().call("san_rss_bwd", _p(_creal(x, "x")), _p(_chk(y, name="y")), _p(_chk(g, name="g")),
                   _p(torch.view_as_real(gx)), n, c, hw, 1, _stream())
    else:
        lib().call("san_rss_bwd", _p(_chk(x, name="x")), _p(_chk(y, name="y")), _p(_chk(g, name="g")), _p(gx), n, c, hw,
                   0, _stream())
    return gx


def _gdev(g: Optional[torch.Tensor]):
    """An upstream gradient scalar kept on the device (autograd's grad_output): fp32, one element."""
    if g is None:
        return None
    assert g.numel() == 1 and g.is_cuda
    return g.detach().reshape(1).to(torch.float32).contiguous()


def ssim_loss_bwd(x: torch.Tensor, y: torch.Tensor, gscale: float = 1.0, gdev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gscale [* gdev] * d ssimloss(x, y) / dy  (gdev: a one-element device tensor, e.g. autograd's grad_output).  SSIM is
    symmetric: d / dx is ssim_loss_bwd(y, x)."""
    n, c, h, w = x.shape
    ws = GLOBAL_ARENA.get("ssim_bwd_ws", (3 * n * (h - 6) * (w - 6),), x.device)
    gy = torch.empty_like(y)
    gd = _gdev(gdev)
    lib().call("san_ssim_loss_bwd_dev", _p(_chk(x, name="x")), _p(_chk(y, name="y")), _p(gy), float(gscale), _p(gd), n, h, w,
               _p(ws), _stream())
    return gy


def lncc_loss_bwd(i: torch.Tensor, j: torch.Tensor, want_i: bool = True, want_j: bool = True, gscale: float = 1.0,
                  gdev: Optional[torch.Tensor] = None, win: int = 9, gi: Optional[torch.Tensor] = None,
                  gj: Optional[torch.Tensor] = None):
    """(gscale [* gdev] * d lncc_loss / d i, ... / d j) (None where not wanted).  Passing gi / gj accumulates into them."""
    _chk(i, name="i")
    _chk(j, name="j")
    n, c, h, w = i.shape
    assert c == 1 and j.shape == i.shape and (want_i or want_j)
    acc = gi is not None or gj is not None
    if not acc:
        gi = torch.empty_like(i) if want_i else None
        gj = torch.empty_like(j) if want_j else None
    ws = GLOBAL_ARENA.get("lncc_bwd_ws", (lib().query("san_lncc_bwd_workspace_floats", n, h, w),), i.device)
    lib().call("san_lncc_loss_bwd", _p(i), _p(j), _p(gi), _p(gj), float(gscale), _p(_gdev(gdev)), int(acc), n, h, w, int(win),
               _p(ws), _stream())
    return gi, gj


def smooth_pool_bwd(gy: torch.Tensor, kern: torch.Tensor, gx: Optional[torch.Tensor] = None, in_hw=None) -> torch.Tensor:
    """Adjoint of smooth_pool: gx [N,1,H,W] (accumulated into when given).  in_hw = the forward input's (H, W) -- needed for
    odd sizes, where avg_pool2d dropped the last row / column (default: 2 x the gradient's size)."""
    _chk(gy, name="gy")
    _chk(kern, name="kern")
    n, c, oh, ow = gy.shape
    acc = gx is not None
    h, w = (int(gx.shape[2]), int(gx.shape[3])) if gx is not None else (in_hw if in_hw is not None else (2 * oh, 2 * ow))
    assert h // 2 == oh and w // 2 == ow, (h, w, oh, ow)
    if gx is None:
        gx = torch.empty((n, c, h, w), device=gy.device, dtype=torch.float32)
    lib().call("san_smooth_pool_bwd", _p(gy), _p(kern), _p(_chk(gx, name="gx")), int(acc), n * c, h, w,
               int(kern.shape[-1]), _stream())
    return gx


def grid_sample_bwd_img(grid: torch.Tensor, g: torch.Tensor, img_shape) -> torch.Tensor:
    """d <g, grid_sample(img, grid)> / d img (zeros padding): a scatter in 64-bit fixed point with integer atomics
    (san_grid_sample_bwd_img_det) -- bit-reproducible, unlike the float-atomic form it replaces (round 5)."""
    n, c, h, w = img_shape
    ho, wo = grid.shape[1:3]
    gimg = torch.empty((n, c, h, w), device=g.device, dtype=torch.float32)
    work = torch.empty((lib().query("san_grid_sample_bwd_img_work_bytes", n, c, h, w) // 8,), device=g.device, dtype=torch.int64)
    lib().call("san_grid_sample_bwd_img_det", _p(_chk(grid, name="grid")), _p(_chk(g, name="g")), _p(gimg), _p(work),
               n, c, h, w, ho, wo, _stream())
    return gimg


def act_bwd_coef(g: Act, y: Act, coef: torch.Tensor, dy: Act) -> None:
    """dy = sc*(u - m1 - (p*yh + q)*m2), coef [n, c, 4] = (m1, m2, p, q)."""
    assert g.c == y.c == dy.c and coef.shape == (y.n, y.c, 4)
    dy.amax = AMAX.next(y.buf.device)
    if dy.amax is None:
        lib().call("san_act_bwd_coef", _p(g.buf), g.ctot, g.coff, _p(y.buf), y.ctot, y.coff, _p(y.scale), _p(y.shift),
                   float(y.slope), _p(_chk(coef, name="coef")), _p(dy.buf), dy.ctot, dy.coff, y.n, y.c, y.h * y.w, _stream())
    else:
        lib().call("san_act_bwd_coef_amax", _p(g.buf), g.ctot, g.coff, _p(y.buf), y.ctot, y.coff, _p(y.scale), _p(y.shift),
                   float(y.slope), _p(_chk(coef, name="coef")), _p(dy.buf), dy.ctot, dy.coff, _p(dy.amax),
                   y.n, y.c, y.h * y.w, _stream())


def warp_bwd_grid(img: torch.Tensor, grid: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """dL/d(offset) NCHW [N,2,H,W] of out = grid_sample(img, grid) given g = dL/d out."""
    n, c, h, w = img.shape
    out = torch.empty((n, 2, h, w), device=img.device, dtype=torch.float32)
    lib().call("san_warp_bwd_grid", _p(_chk(img, name="img")), _p(_chk(grid, name="grid")), _p(_chk(g, name="g")),
               _p(out), n, c, h, w, _stream())
    return out


def gradient_loss_bwd(offset_nchw: torch.Tensor, g: torch.Tensor, gscale: float, accumulate: bool,
                      gdev: Optional[torch.Tensor] = None) -> None:
    n, two, h, w = offset_nchw.shape
    lib().call("san_gradient_loss_bwd_dev", _p(_chk(offset_nchw, name="offset")), _p(_chk(g, name="g")), float(gscale),
               _p(_gdev(gdev)), int(accumulate), n, h, w, _stream())
