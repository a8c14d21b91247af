"""ctypes binding of libsan_hip.so (the C ABI declared in include/san_hip.h).

The prototypes are parsed from the header itself, so the binding cannot drift
from the ABI and a test can check that every declared symbol is exported.
There is NO fallback: if the shared object is missing or a symbol is absent the
import raises, and every product entry point goes through this module.
"""
from __future__ import annotations

import ctypes
import os
import re
import struct
from typing import Dict, List, Tuple

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime libsan_hip.so links against)

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "san_hip.h")
LIB_PATH = os.environ.get("SAN_LIB_PATH") or os.path.join(HERE, "libsan_hip.so")     # SAN_LIB_PATH: same-box A/B of two builds

# ---------------------------------------------------------------------------------------------------------------
# Step recorder.  A training step issues the SAME ~2,000 C-ABI calls with the same arguments every time once the arenas are
# warm; what differs between steps lives in device memory.  While REC is a list, every C-ABI call, every stream / event
# operation and every torch operation of the step that is routed through ``rec()`` is executed AND remembered as
# ``(callable, args)``; ``CSModel.record_update`` turns one recorded step into an object whose ``replay()`` is a flat loop
# over those pairs -- the Python between the calls (layer logic, arena look-ups, argument marshalling decisions: three
# quarters of the ~45 ms a step costs the host) is not run again.  Unlike a hipGraph this keeps ordinary stream semantics (and
# ROCm's graph launch was measured to cost the host as much as eager launching).
REC = None          # list of (callable, args, kind) while recording; kind 0 = Python callable, 1 = C-ABI call (returns an rc that a
                    # replay checks), 2 = C-ABI weight-packing launch (a forward-only replay skips it while the weights are unchanged)
KEEP = None         # objects whose addresses were recorded (host scratch of C calls)
IN_REC = [0]        # > 0 inside rec(): torch operations seen by the recording's dispatch mode are accounted for
UNTRACKED = [0]     # > 0 inside untracked(): torch operations that need no replay (constants, host-side scalars)
AUX = [None]        # the stream of the auxiliary region being recorded (ops.aux_region): a torch callable recorded inside it is
                    # replayed with that stream current (a replay runs with the main stream current)


def rec(fn, *args):
    """Run ``fn(*args)`` now; remember it when a step is being recorded."""
    IN_REC[0] += 1
    try:
        out = fn(*args)
    finally:
        IN_REC[0] -= 1
    if REC is not None:
        aux = AUX[0]
        if aux is not None and getattr(fn, "__self__", None).__class__.__name__ not in ("Stream", "Event", "ExternalStream"):
            REC.append((_under_stream(aux, fn), args, 0))       # (stream / event methods name their streams themselves)
        else:
            REC.append((fn, args, 0))
    return out


def _under_stream(stream, fn):
    def run(*args):
        with torch.cuda.stream(stream):
            return fn(*args)
    run.__name__ = getattr(fn, "__name__", "callable") + "@aux"
    return run


class untracked:
    """``with untracked():`` -- torch operations inside do not belong to the replayed step (constant results such as the
    float copy of the sampling mask, or host-visible scalars nobody replays such as ``loss_all``)."""

    def __enter__(self):
        UNTRACKED[0] += 1

    def __exit__(self, *exc):
        UNTRACKED[0] -= 1
        return False


_CT = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """{function name: (restype, [argtypes])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int|size_t)\s+(san_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else _CT[ret]
        argtypes = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CT[a.split()[0] if a.split()[0] != "const" else a.split()[1]])
        protos[name] = (restype, argtypes)
    return protos


class SanLibrary:
    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -m spatialalignmentnetwork_amd.build` "
                "(there is no CPU or PyTorch fallback for the hot path)")
        self._memo = {}
        self._fns = {}
        self._setters = set()
        self._dll = ctypes.CDLL(path)
        self.protos = parse_header()
        self.func_index = {name: i for i, name in enumerate(self.protos)}      # = the case labels of csrc/san_replay_table.inc
        for name, (restype, argtypes) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:  # pragma: no cover
                raise RuntimeError(f"libsan_hip.so does not export {name}") from e
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, "_" + name, fn)

    def last_error(self) -> str:
        return self._san_last_error_string().decode()

    # size / eligibility / geometry queries are pure functions of their integer arguments (and of the tuning hooks, whose
    # setters clear the cache): memoised, ~2,800 of them per training step otherwise cross ctypes
    _STATEFUL = frozenset({"san_get_conv_precision", "san_wgrad_defer", "san_wgrad_defer_pending", "san_version",
                           "san_conv_direct_enable", "san_conv1x1_gemm_enable", "san_conv_f16_wscale_enable"})

    def call(self, name: str, *args):
        """Call an int-returning entry point; raise RuntimeError on failure.  (~2,000 calls per training step: the function
        object comes from a dict, and only the few setters pay for clearing the query cache.)"""
        try:
            fn = self._fns[name]
        except KeyError:
            fn = self._fns[name] = getattr(self, "_" + name)
            if name.startswith("san_set_") or name.endswith("_set_tuning"):
                self._setters.add(name)
        if name in self._setters:
            self._memo.clear()
        rc = fn(*args)
        if rc != 0:
            kind = "argument error" if rc < 0 else "hipError_t"
            raise RuntimeError(f"{name} failed ({kind} {rc}): {self.last_error()}")
        if REC is not None:
            REC.append((fn, args, 2 if "_pack" in name else 1))

    def query(self, name: str, *args):
        """Call a size/count query (returns its value)."""
        if name in self._STATEFUL:
            fn = getattr(self, "_" + name)
            if REC is not None and name == "san_wgrad_defer":      # switches the library's deferred-reduction mode: part of the step
                REC.append((fn, args, 0))         # (returns the previous mode, not an rc)
            return fn(*args)
        key = (name, args)
        try:
            return self._memo[key]
        except KeyError:
            v = self._memo[key] = getattr(self, "_" + name)(*args)
            return v
        except TypeError:                       # an unhashable argument (a ctypes object): not a geometry query
            return getattr(self, "_" + name)(*args)


# ---------------------------------------------------------------------------------------------------------------
# Tapes for san_replay_run (csrc/san_replay.cpp): a recorded step's C-ABI calls and stream / event operations as 64-bit words
# that ONE foreign call walks in C, instead of ~4,000 ctypes calls (~10 us of argument marshalling each) per step.
TAPE_EVENT_RECORD, TAPE_STREAM_WAIT = 0x8000, 0x8001
TAPE_IGNORE_RC, TAPE_PACK = 1, 2
_F32 = struct.Struct("<f")
_U32 = struct.Struct("<I")
_F64 = struct.Struct("<d")
_U64 = struct.Struct("<Q")


def tape_call_words(fn, args, flags: int = 0):
    """The tape entry of the C-ABI call ``fn(*args)`` (fn: a function object of SanLibrary), or None when it cannot be put on a
    tape (a char*-returning query, an argument that is not a plain value)."""
    protos = lib().protos
    name = getattr(fn, "__name__", None)
    if name not in protos or protos[name][0] is not ctypes.c_int or name == "san_replay_run":
        return None
    argtypes = protos[name][1]
    if len(args) != len(argtypes) or len(args) > 255:
        return None
    words = [lib().func_index[name] | flags << 16 | len(args) << 24]
    for a, t in zip(args, argtypes):
        if t is ctypes.c_void_p:
            if a is None:
                v = 0
            elif isinstance(a, int):
                v = a
            elif isinstance(a, ctypes.c_void_p):
                v = a.value or 0
            else:
                return None
        elif t is ctypes.c_int:
            v = int(a) & 0xFFFFFFFFFFFFFFFF       # (sign-extended two's complement: the dispatcher narrows it back to int)
        elif t is ctypes.c_size_t:
            v = int(a)
        elif t is ctypes.c_float:
            v = _U32.unpack(_F32.pack(float(a)))[0]
        elif t is ctypes.c_double:
            v = _U64.unpack(_F64.pack(float(a)))[0]
        else:
            return None
        if not 0 <= v <= 0xFFFFFFFFFFFFFFFF:
            return None
        words.append(v)
    return words


# Events of a recorded step that only order this GPU's own streams (weight-gradient hand-offs, wait_stream): a replay tape gives them
# RAW HIP events created without the system-scope fence (hipEventDisableSystemFence): the record then costs the stream ~1.2 us instead
# of ~3.2 us (scratch/r5_event_flags.py; agent scope is all a same-device dependency needs).  ``light(ev)`` marks a torch event as such
# while a step is recorded; timing events and anything the host reads stay torch's.
LIGHT = None        # set of id(torch event) while recording


def light(ev):
    if REC is not None and LIGHT is not None:
        LIGHT.add(id(ev))
    return ev


class RawEvents:
    """hipEvent_t handles without timing, owned by a recorded step's tapes, made by libsan_hip.so itself (san_event_create:
    the runtime the library links, on the device of the stream that will record them -- round 6).  ``new(device,
    light)``: light drops the system-scope fence (same-GPU hand-offs only)."""

    def __init__(self):
        self.handles = []

    def new(self, device: int, light: bool = True) -> int:
        e = ctypes.c_void_p()
        lib().call("san_event_create", int(device), 1 if light else 0, ctypes.byref(e))
        if not e.value:
            raise RuntimeError("san_event_create returned no handle")
        self.handles.append(e.value)
        return e.value

    def __del__(self):
        try:
            for h in self.handles:
                lib()._san_event_destroy(ctypes.c_void_p(h))
        except Exception:
            pass


class Tape:
    """A finished tape: ``run(skip_packs)`` walks it; a failing entry raises with the recorded call's name."""

    def __init__(self, words, names):
        self.n = len(words)
        self.buf = (ctypes.c_uint64 * max(self.n, 1))(*words)
        self.names = names                      # {head word index: what the entry is} for error messages
        self._where = ctypes.c_longlong(-1)
        self._run = lib()._san_replay_run
        self._args = (ctypes.addressof(self.buf), self.n)
        self._wp = ctypes.addressof(self._where)

    def run(self, skip_packs: bool = False) -> None:
        rc = self._run(self._args[0], self._args[1], 1 if skip_packs else 0, self._wp)
        if rc:
            what = self.names.get(self._where.value, "?")
            raise RuntimeError(f"recorded step: {what} failed ({'argument error' if rc < 0 else 'hipError_t'} {rc}): {lib().last_error()}")


_LIB = None


def lib() -> SanLibrary:
    global _LIB
    if _LIB is None:
        _LIB = SanLibrary()
    return _LIB
