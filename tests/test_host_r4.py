"""Round-4 host-side checks that need no GPU: the ADVICE r3 items (legacy Config checkpoints through the safe loader, the
memoised trainable-parameter list noticing replaced parameters, separate event rings, backend detection for a process group the
user initialised) and the recorder's bookkeeping."""
import io
import sys
import types

import torch

from spatialalignmentnetwork_amd import _lib, autograd, basemodel, dist as sdist, ops


def test_legacy_config_pickle_loads_through_the_safe_loader(tmp_path):
    """The reference pickles ``basemodel.Config`` objects into its torch checkpoints (basemodel.py:17-41,57): the safe loader
    must build them without SAN_TRUST_CHECKPOINTS."""
    mod = types.ModuleType("basemodel")

    def _init(self):
        self.memo = ["sparsity", "shape"]
        self.sparsity, self.shape = 0.25, 320

    Config = type("Config", (object,), {"__init__": _init, "__module__": "basemodel", "__qualname__": "Config"})
    mod.Config = Config
    sys.modules["basemodel"] = mod
    try:
        buf = io.BytesIO()
        torch.save({"config": Config(), "net_R": {"w": torch.arange(4.0)}}, buf)
    finally:
        del sys.modules["basemodel"]
    path = tmp_path / "legacy.pt"
    path.write_bytes(buf.getvalue())
    out = basemodel._torch_load(str(path))
    assert isinstance(out["config"], basemodel.Config) and out["config"].shape == 320 and "sparsity" in out["config"]
    assert torch.equal(out["net_R"]["w"], torch.arange(4.0))


def test_trainable_cache_notices_replaced_parameters_and_submodules():
    m = torch.nn.Sequential(torch.nn.Linear(3, 3), torch.nn.Linear(3, 2))
    a = autograd._trainable(m)
    assert len(a) == 4
    m[0].weight.requires_grad_(False)
    assert len(autograd._trainable(m)) == 3
    new = torch.nn.Parameter(torch.zeros(3, 3))
    m[0].weight = new                                   # replacement: the old object must not be handed out any more
    assert any(p is new for p in autograd._trainable(m))
    m[1] = torch.nn.Linear(3, 5)                        # swapped submodule
    got = autograd._trainable(m)
    assert any(p is m[1].weight for p in got) and len(got) == 4


def test_side_stream_marks_have_their_own_event_ring():
    assert ops._EVENTS is not ops._BUSY_EVENTS and ops._EVENTS.ev is not ops._BUSY_EVENTS.ev


def test_backend_falls_back_to_the_process_group(monkeypatch):
    import torch.distributed as dist
    monkeypatch.setattr(sdist, "BACKEND", None)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_backend", lambda *a, **k: "gloo")
    assert sdist.backend() == "gloo"
    monkeypatch.setattr(sdist, "BACKEND", "nccl")
    assert sdist.backend() == "nccl"


def test_recorder_tags_packing_launches_and_python_callables():
    """_lib.rec entries are Python callables (kind 0); C-ABI calls carry kind 1, weight-packing launches kind 2 (a forward-only
    replay skips them while the weights are unchanged) -- without a GPU only the bookkeeping is checked."""
    seen = []
    _lib.REC = []
    try:
        _lib.rec(seen.append, 7)
        assert _lib.REC == [(seen.append, (7,), 0)] and seen == [7]
    finally:
        _lib.REC = None
    assert "_pack" in "san_conv_bf16x3_pack_batch" and "_pack" in "san_conv_pack_batch" and "_pack" not in "san_conv2d_bf16x3_fwd"
