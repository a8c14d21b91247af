"""Data-parallel plumbing for the slice-sharded path (one process per GPU).

Slices are independent (every norm on the VarNet path is per sample; BatchNorm in
the alignment net keeps per-replica statistics, as the reference does), so
inference needs NO data-path collective: each rank takes a contiguous shard of
the global batch.  Only scalars (timing, metrics) cross ranks.  Backend is
"nccl" (= RCCL over xGMI) on GPUs and "gloo" in the CPU tests.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


BACKEND = None   # the backend actually in use (bench.py reports it)


def init(backend: str, device=None, allow_fallback: bool = False):
    """Initialise torch.distributed if WORLD_SIZE > 1; returns the module or None.
    backend "nccl" is RCCL on ROCm.  RCCL is probed with one all-reduce so that a broken transport fails HERE and not
    inside a timed region.  A failure raises: a multi-GPU number whose gradients went through the host over gloo would
    be meaningless.  Only ``allow_fallback=True`` (debugging on a box without working IPC) degrades to gloo, with the
    gradients staged through the host; ``BACKEND`` records what is in use."""
    global BACKEND
    rank, _, world = env_rank_world()
    if world <= 1 and not single_rank_exchange():
        return None
    import torch.distributed as dist
    if backend == "nccl" and device is not None:
        prebind_streams(device)
    if world <= 1:                                 # (the hook below: a one-rank RCCL communicator on this GPU)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        try:
            kw = {"device_id": device} if device is not None else {}
            dist.init_process_group("nccl", **kw)
            t = torch.zeros(1, device=device)
            dist.all_reduce(t)                      # fail here, not inside the timed region
            torch.cuda.synchronize()
            BACKEND = "nccl"
        except Exception as e:                     # pragma: no cover (needs a multi-GPU box)
            if not allow_fallback:
                raise RuntimeError(f"RCCL could not initialise ({type(e).__name__}: {e}); refusing to fall back to gloo "
                                   "(pass allow_fallback=True to stage gradients through the host)") from e
            print(f"[dist] RCCL unavailable ({type(e).__name__}: {e}); falling back to gloo", flush=True)
            try:
                dist.destroy_process_group()
            except Exception:
                pass
            backend = "gloo"
        else:
            if device is not None:
                # the step's own communicator (collective: every rank is here).  Its failure only switches the native path off
                # -- nccl itself came up, so nothing here may send the job to gloo
                try:
                    native_rccl(dist, device)
                except Exception as e:
                    NATIVE.update(handle=None, tried=True, key=_group_key(dist), why=f"{type(e).__name__}: {e}")
            return dist
    dist.init_process_group(backend)
    BACKEND = backend
    return dist


def shutdown() -> None:
    """Release the package's RCCL communicator and forget the backend (call before ``destroy_process_group``; a later ``init``
    builds a new communicator for the new group)."""
    global BACKEND
    h = NATIVE.get("handle")
    if h is not None:
        try:
            from . import _lib
            _lib.lib().call("san_rccl_comm_destroy", int(h))
        except Exception:
            pass
    NATIVE.update(handle=None, tried=False, key=None, why="not tried")
    BACKEND = None


# ---------------------------------------------------------------------------------------------------------------
# Native RCCL communicator for the step's all-reduces (csrc/san_rccl.cpp, round 5): a recorded step then holds its collectives
# as C-ABI tape entries (san_rccl_allreduce_sum_f32 on the communication stream) instead of Python closures around
# torch.distributed.all_reduce.  torch's process group stays in charge of everything else (and of the step itself whenever the
# native communicator cannot be built: SAN_NATIVE_RCCL=0, a backend other than nccl, a failed or timed-out ncclCommInitRank on
# ANY rank -- the ranks agree on the outcome through the process group before anybody uses it).
NATIVE = {"handle": None, "tried": False, "why": "not tried", "version": None, "key": None}


def _group_key(dist):
    """What a cached communicator belongs to: the default process group OBJECT (a re-initialised group is a new one), its
    world size and this rank, and the backend."""
    try:
        if dist is None or not dist.is_initialized():
            return None
    except Exception:                   # (not a torch.distributed module: nothing to key on)
        return None
    try:
        from torch.distributed import distributed_c10d as c10d
        gid = id(c10d._get_default_group())
    except Exception:
        gid = 0
    return (gid, dist.get_world_size(), dist.get_rank(), backend())


def _loaded_rccl_path():
    try:
        for line in open("/proc/self/maps"):
            if "librccl" in line:
                return line.split()[-1]
    except OSError:
        pass
    return None


def native_rccl(dist, device, timeout_s: float = 120.0):
    """Handle of the package's own RCCL communicator over the ranks of ``dist`` (built on first use: a COLLECTIVE call), or None."""
    key = _group_key(dist)
    if NATIVE["tried"] and NATIVE["key"] == key:
        return NATIVE["handle"]
    from . import _lib
    if _lib.REC is not None or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return None                     # (never built inside a recording / capture: the eager warm-up steps come first)
    if NATIVE["handle"] is not None:    # the process group changed under a live communicator: it belongs to the old ranks
        try:
            _lib.lib().call("san_rccl_comm_destroy", int(NATIVE["handle"]))
        except Exception:
            pass
    NATIVE.update(handle=None, tried=True, key=key, why="not tried")
    if os.environ.get("SAN_NATIVE_RCCL", "1") == "0":
        NATIVE["why"] = "SAN_NATIVE_RCCL=0"
        return None
    if dist is None or backend() != "nccl" or device is None or torch.device(device).type != "cuda":
        NATIVE["why"] = f"backend {backend()}"
        return None
    import ctypes
    import threading
    lib = _lib.lib()
    device = torch.device(device)
    world, rank = dist.get_world_size(), dist.get_rank()
    ok, why, handle = 1, "", ctypes.c_int(-1)
    idbuf = torch.zeros(128, dtype=torch.uint8)
    try:
        ver = ctypes.c_int(0)
        lib.call("san_rccl_load", (_loaded_rccl_path() or "").encode(), ctypes.byref(ver))
        NATIVE["version"] = ver.value
        if rank == 0:
            raw = ctypes.create_string_buffer(128)
            lib.call("san_rccl_unique_id", raw)
            idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
    except Exception as e:                                  # (every rank still takes part in the exchanges below)
        ok, why = 0, f"{type(e).__name__}: {e}"
    dev_id = idbuf.to(device)
    dist.broadcast(dev_id, src=0)
    idbytes = bytes(dev_id.cpu().numpy().tobytes())
    if ok:
        res = {}

        def _init():
            try:
                torch.cuda.set_device(device)               # (the HIP device is per thread)
                lib.call("san_rccl_comm_init", idbytes, world, rank, ctypes.byref(handle))
                res["ok"] = True
            except Exception as e:
                res["err"] = f"{type(e).__name__}: {e}"

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            ok, why = 0, f"ncclCommInitRank did not return within {timeout_s:.0f} s"
        elif not res.get("ok"):
            ok, why = 0, res.get("err", "ncclCommInitRank failed")
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) and ok:
        # one probe sum through the new communicator, checked and agreed on, before the step depends on it
        probe = torch.ones(4, dtype=torch.float32, device=device)
        try:
            lib.call("san_rccl_allreduce_sum_f32", handle.value, probe.data_ptr(), 4, torch.cuda.current_stream(device).cuda_stream)
            torch.cuda.synchronize(device)
            good = bool((probe == float(world)).all().item())
        except Exception as e:
            good, why = False, f"{type(e).__name__}: {e}"
        flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            NATIVE["handle"], NATIVE["why"] = handle.value, "ok"
            return handle.value
        why = why or "probe all-reduce gave a wrong sum on some rank"
    NATIVE["why"] = why or "another rank could not build the communicator"
    return None


def prebind_streams(device) -> None:
    """Create the step's own streams (weight gradients, gradient exchange) and run one kernel on each BEFORE RCCL comes up.  HIP
    binds a stream to one of its few hardware queues at first use; with the process group initialised first (what bench.py and
    train.py do) RCCL's streams took them and the weight-gradient stream ended up sharing a queue with the main stream -- no
    overlap left: 49.9 instead of 45.5 ms per step with the exchange not even running (round 4, scratch/rccl1_where2.py).
    ``init()`` calls it; a program that calls ``torch.distributed.init_process_group`` itself should call it first."""
    from . import ops
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    dev = torch.device("cuda", idx)
    side = ops._WG["pool"].get(idx)
    if side is None:
        side = ops._WG["pool"][idx] = torch.cuda.Stream(device=dev)
    comm = GradExchange._streams.get(str(dev))
    if comm is None:
        comm = GradExchange._streams[str(dev)] = torch.cuda.Stream(device=dev)
    for st in (torch.cuda.current_stream(dev), side, comm):
        with torch.cuda.stream(st):
            torch.zeros(64, device=dev).add_(1.0)
    torch.cuda.synchronize(dev)


def exchange_mode() -> str:
    """``SAN_GRAD_EXCHANGE``: "allreduce" (default: one ncclAllReduce per slice) or "rs_ag" (reduce-scatter + all-gather per slice;
    native RCCL communicator only -- anything else falls back to the all-reduce)."""
    m = os.environ.get("SAN_GRAD_EXCHANGE", "allreduce")
    if m not in ("allreduce", "rs_ag"):
        raise ValueError(f"SAN_GRAD_EXCHANGE={m!r}: choose allreduce or rs_ag")
    return m


def agree_on_failure(err, dist, device):
    """Every rank passes its own failure (a message, or None); all of them get a message back if ANY rank failed (their own, or
    "another rank ...") and None only if every rank succeeded: the ranks then take the same branch (CSModel.update: replay the
    recorded step everywhere or stay eager everywhere -- a rank that failed half-way must not be left alone with the others'
    next collective)."""
    if dist is None:
        return err
    flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=_scalar_device(device))
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) and err is None:
        err = "another rank could not record the step"
    return err


def single_rank_exchange() -> bool:
    """``SAN_DIST_SINGLE=1``: treat a ONE-rank process group as data-parallel, i.e. run the whole gradient exchange (RCCL
    communicator, communication stream, per-cascade slices, recorded / captured collectives) with world size 1.  A one-GPU box
    cannot show the transport, but it does run every RCCL call site of the step (tests/test_gpu_dist.py,
    ``bench.py`` with the variable set); sums over one rank leave the gradients unchanged, so the step must equal the plain one
    bit for bit."""
    return os.environ.get("SAN_DIST_SINGLE", "0") == "1"


def backend() -> str:
    """The backend in use: what init() recorded, else what a process group initialised directly through torch.distributed
    (the reference's workflow) reports."""
    if BACKEND is not None:
        return BACKEND
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return str(dist.get_backend())
    return None


def _scalar_device(device):
    return device if backend() == "nccl" else "cpu"


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of `total` slices owned by `rank`: contiguous, sizes differ by at most one,
    every slice owned exactly once."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, dist, device="cpu") -> float:
    """The slowest rank's time (bench.py's max-over-ranks rule)."""
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_scalar_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value: float, dist, device="cpu"):
    """[rank 0's value, rank 1's, ...] on every rank (bench.py: per-rank exchange timings)."""
    if dist is None:
        return [float(value)]
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=_scalar_device(device))
    t[dist.get_rank()] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)        # (one-hot rows: a sum is a gather that every backend has)
    return [float(v) for v in t.tolist()]


def sum_over_ranks(value: float, dist, device="cpu") -> float:
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_scalar_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_metric_mean(local_sum: float, local_count: int, dist, device="cpu") -> float:
    """Global mean of a per-slice metric from per-rank (sum, count)."""
    s = sum_over_ranks(local_sum, dist, device)
    c = sum_over_ranks(float(local_count), dist, device)
    return s / max(c, 1.0)


def broadcast0(t: torch.Tensor, dist) -> None:
    """In-place broadcast of rank 0's tensor (bool buffers travel as uint8; CUDA tensors are staged through the host
    when only gloo is up)."""
    if dist is None:
        return
    src = t
    if t.dtype == torch.bool:
        src = t.to(torch.uint8)
    if src.is_cuda and backend() == "gloo":
        host = src.cpu()
        dist.broadcast(host, src=0)
        src = host.to(t.device)
    else:
        if not src.is_contiguous():
            src = src.contiguous()
        dist.broadcast(src, src=0)
    if src is not t:
        t.copy_(src.to(t.dtype))


class GradBucket:
    """One flat fp32 gradient buffer for a set of parameters; every ``p.grad`` is a view into it, so
    the data-parallel exchange needs no packing copies: the whole buffer in one in-place all-reduce, or -- ``range_of`` --
    contiguous slices of it (one per cascade, in reverse order, as their gradients become final: GradExchange.launch)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.offsets = []
        off = 0
        for p in self.params:                      # every tensor starts on a 16-byte boundary
            self.offsets.append(off)
            off += (p.numel() + 3) & ~3
        self.total = off
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def zero(self):
        from . import _lib
        _lib.rec(self.flat.zero_)

    def range_of(self, params):
        """[lo, hi) of the flat buffer that holds the gradients of ``params`` (they must be consecutive in this bucket's
        order, as the parameters of one submodule are); None if none of them is in the bucket."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not idx:
            return None
        if idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError("the parameters are not consecutive in the bucket")
        last = idx[-1]
        return self.offsets[idx[0]], self.offsets[last] + ((self.params[last].numel() + 3) & ~3)

    def allreduce_sum(self, dist, rng=None) -> None:
        """Sum the gradients over all ranks in place (the optimiser kernel applies the 1/world factor); rng = (lo, hi):
        that slice of the flat buffer only."""
        if dist is None:
            return
        flat = self.flat if rng is None else self.flat[rng[0]:rng[1]]
        if flat.is_cuda and backend() == "gloo":      # RCCL unavailable: stage through the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)

    def allreduce_mean(self, dist) -> None:
        """Average the gradients over all ranks (sum, then divide by the world size)."""
        if dist is None:
            return
        self.allreduce_sum(dist)
        self.flat.div_(dist.get_world_size())


class GradExchange:
    """The data-parallel gradient exchange of one training step, off the main stream.

    ``launch(bucket, after, rng)`` starts the summing all-reduce of a flat gradient buffer -- or of the slice ``rng`` of it --
    on a communication stream as soon as its producers are done (the main stream up to the call + the streams in ``after``,
    i.e. the weight-gradient side stream); ``wait()`` joins everything back into the main stream in front of the optimiser.
    net_R's 119.8 MB go out cascade by cascade in reverse order (~9.8 MB each) from inside ``VarNet.backward``
    and hide behind the remaining cascades' backward; the sensitivity net's slice and net_T's 2.9 MB follow.
    RCCL collectives are stream-ordered, so both calls are capturable into a hipGraph.  With gloo (CPU tests, debugging)
    the buffer is staged through the host synchronously -- same call sites, no overlap."""

    _streams = {}

    def __init__(self, dist, timed: bool = False):
        self.dist = dist
        self.works = []
        self.launched = []              # the slices (None = a whole buffer) in launch order
        self.timed = timed
        self.events = []
        self.wait_events = []           # (before, after) the main stream's join: how long the step WAITED for the exchange
        self.comm = None
        self.mode = exchange_mode()

    def _comm(self, device):
        key = str(device)
        if key not in GradExchange._streams:
            GradExchange._streams[key] = torch.cuda.Stream(device=device)
        return GradExchange._streams[key]

    def launch(self, bucket: "GradBucket", after=(), rng=None) -> None:
        if self.dist is None:
            return
        flat = bucket.flat if rng is None else bucket.flat[rng[0]:rng[1]]
        self.launched.append(rng)
        from . import _lib
        if not flat.is_cuda or backend() == "gloo":
            if flat.is_cuda:
                cur = torch.cuda.current_stream()
                for s in after:
                    if s is not None:
                        _lib.rec(cur.wait_stream, s)
            _lib.rec(bucket.allreduce_sum, self.dist, rng)
            return
        cur = torch.cuda.current_stream()
        comm = self.comm = self._comm(flat.device)
        _lib.rec(comm.wait_stream, cur)
        for s in after:
            if s is not None:
                _lib.rec(comm.wait_stream, s)
        timed = self.timed and not torch.cuda.is_current_stream_capturing()
        dist = self.dist

        def _collective():
            with torch.cuda.stream(comm):
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
                work.wait()                     # stream-level: the communication stream waits for the collective

        e0 = e1 = None
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _lib.rec(e0.record, comm)
        nat = native_rccl(dist, flat.device)
        if nat is not None:
            # C-ABI calls on the communication stream: tape entries of the recorded step, no Python at replay time
            world, rank = dist.get_world_size(), dist.get_rank()
            chunk = flat.numel() // world if self.mode == "rs_ag" else 0
            if chunk:
                # SAN_GRAD_EXCHANGE=rs_ag: the sum as a direct reduce-scatter + all-gather, both in place (rank r owns
                # chunk r): on a fully connected xGMI node each phase moves 1/world of the slice over every link at once.  The few
                # elements past world * chunk take a small all-reduce.  Deterministic, but a different summation order than
                # ncclAllReduce's for more than two ranks: all ranks of a job use the same mode (part of the recording's key).
                base, body = flat.data_ptr(), chunk * world
                _lib.lib().call("san_rccl_reduce_scatter_sum_f32", nat, base, base + 4 * rank * chunk, chunk, comm.cuda_stream)
                _lib.lib().call("san_rccl_allgather_f32", nat, base + 4 * rank * chunk, base, chunk, comm.cuda_stream)
                if flat.numel() > body:
                    _lib.lib().call("san_rccl_allreduce_sum_f32", nat, base + 4 * body, flat.numel() - body, comm.cuda_stream)
            else:
                _lib.lib().call("san_rccl_allreduce_sum_f32", nat, flat.data_ptr(), flat.numel(), comm.cuda_stream)
        else:
            _lib.rec(_collective)
        if timed:
            _lib.rec(e1.record, comm)
            self.events.append((e0, e1))
        flat.record_stream(comm)

    def wait(self) -> None:
        if self.comm is not None:
            from . import _lib
            cur = torch.cuda.current_stream()
            timed = self.timed and not torch.cuda.is_current_stream_capturing()
            if timed:
                # an event pair around the join on the MAIN stream: what it measures is the exchange the backward did not hide
                w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                _lib.rec(w0.record, cur)
            _lib.rec(cur.wait_stream, self.comm)
            if timed:
                _lib.rec(w1.record, cur)
                self.wait_events.append((w0, w1))
        self.works.clear()

    def elapsed_ms(self) -> float:
        """Sum of the collectives' durations on the communication stream (needs a prior synchronisation)."""
        return float(sum(e0.elapsed_time(e1) for e0, e1 in self.events))

    def exposed_ms(self) -> float:
        """How long the main stream waited for the exchange at the join (needs a prior synchronisation)."""
        return float(sum(e0.elapsed_time(e1) for e0, e1 in self.wait_events))


class ParamBucket(GradBucket):
    """GradBucket plus flat parameter and AdamW moment buffers: every ``p.data`` becomes a view into
    ``flat_p`` (same Parameter objects, same state_dict keys and shapes), so one fused kernel
    (san_adamw_step) updates a whole network."""

    def __init__(self, params):
        super().__init__(params)
        self.flat_p = torch.zeros_like(self.flat)
        for p, o in zip(self.params, self.offsets):
            view = self.flat_p[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = 0
        self.step_dev = None            # int64 [1] on the device once FusedAdamW runs in its graph-safe form

    def owns(self, params) -> bool:
        """True while every parameter still lives in this bucket (``module.to()`` re-allocates them)."""
        lo = self.flat_p.data_ptr()
        hi = lo + self.flat_p.numel() * 4
        ps = [p for p in params if p.requires_grad]
        return len(ps) == len(self.params) and all(
            a is b and lo <= a.data_ptr() < hi and a.grad is not None for a, b in zip(ps, self.params))
