"""Host-side mirror of the reference's basemodel.py: Config bag, directory
checkpoints (JSON ``config`` + one extension-less ``np.savez`` blob per network,
keyed by state_dict names) and the attribute-introspecting BaseModel.
Reference: basemodel.py:17-182.  No kernels here; the on-disk format is kept so
directories written by the reference load unchanged."""
from __future__ import annotations

import json
import os

import numpy as np
import torch


class Config(object):
    """Attribute bag that remembers insertion order; JSON save/load.  basemodel.py:57-100."""

    def __init__(self, **params):
        object.__setattr__(self, "memo", [])
        for k, v in params.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if name not in self.memo:
            self.memo.append(name)
        object.__setattr__(self, name, value)

    def __delattr__(self, name):
        self.memo.remove(name)
        object.__delattr__(self, name)

    def __getitem__(self, name):
        assert name in self.memo, f"{name} not found, try {self.memo}"
        return getattr(self, name)

    def __contains__(self, name):
        return name in self.memo

    def __repr__(self):
        return "class Config containing: " + str({k: getattr(self, k) for k in self.memo})

    __str__ = __repr__

    def save(self, path):
        with open(path, "w") as f:
            json.dump({k: getattr(self, k) for k in self.memo}, f)

    def load(self, path):
        for k in list(self.memo):
            delattr(self, k)
        with open(path, "r") as f:
            for k, v in json.load(f).items():
                setattr(self, k, v)


class _LegacyConfig(Config):
    """``basemodel.Config`` as the reference's torch-pickled checkpoints name it (allow-listed for the safe loader)."""


_LegacyConfig.__module__, _LegacyConfig.__name__, _LegacyConfig.__qualname__ = "basemodel", "Config", "Config"


def ckpt_save(ckpt: dict, folder: str) -> None:
    """basemodel.py:43-55."""
    assert isinstance(ckpt, dict)
    assert not os.path.exists(folder), folder + " already exists"
    os.mkdir(folder)
    for key, val in ckpt.items():
        path = os.path.join(folder, key)
        if key == "config":
            val.save(path)
        else:
            with open(path, "wb") as f:
                np.savez(f, **{k: v.detach().cpu().numpy() for k, v in val.items()})


def _torch_load(path: str):
    """Legacy torch-pickle checkpoints (basemodel.py:17-41 accepts them).  Tensors / state_dicts load with the safe
    unpickler (weights_only=True); a pickled ``Config`` object is allow-listed for it.  Anything else the safe loader
    rejects is NOT retried with the full unpickler (which executes code from the file) unless the caller opts in with
    SAN_TRUST_CHECKPOINTS=1 -- and then with a warning that names the first error."""
    import warnings
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as first:                  # (pickle.UnpicklingError for a refused global; other loader errors take the same road)
        try:
            # the reference pickles its Config under the global 'basemodel.Config' (basemodel.py:57): a subclass that carries
            # that module / name is allow-listed next to this package's own, so the safe loader builds it
            with torch.serialization.safe_globals([Config, _LegacyConfig]):
                return torch.load(path, map_location="cpu", weights_only=True)
        except Exception:
            pass
        if os.environ.get("SAN_TRUST_CHECKPOINTS", "0") != "1":
            raise RuntimeError(f"{path}: the safe loader refused this checkpoint ({first}); set SAN_TRUST_CHECKPOINTS=1 to "
                               "unpickle it with full code execution if you trust its origin") from first
        warnings.warn(f"{path}: unpickling with full code execution (SAN_TRUST_CHECKPOINTS=1); safe loader said: {first}")
        return torch.load(path, map_location="cpu", weights_only=False)


def ckpt_load(folder: str) -> dict:
    """basemodel.py:17-41: a directory of npz blobs (or torch pickles), or a single torch file."""
    if os.path.isfile(folder):
        return _torch_load(folder)
    ckpt = {}
    for key in sorted(os.listdir(folder)):
        path = os.path.join(folder, key)
        if key == "config":
            cfg = Config()
            try:
                cfg.load(path)
            except (UnicodeDecodeError, json.JSONDecodeError):
                cfg = _torch_load(path)
            ckpt[key] = cfg
            continue
        try:
            with np.load(path) as z:
                ckpt[key] = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
        except Exception:
            ckpt[key] = _torch_load(path)
    return ckpt


class BaseModel(object):
    """basemodel.py:102-182: holds nn.Modules as attributes; to/train/eval/save/load by introspection."""

    def __init__(self, cfg=None, ckpt=None, objects=None):
        if ckpt is not None:
            self.load(cfg=cfg, ckpt=ckpt, objects=objects)
        else:
            self.build(cfg)
        self.training = True

    def build(self, cfg):
        self.cfg = cfg

    def _modules(self):
        return {k: v for k, v in self.__dict__.items() if isinstance(v, torch.nn.Module)}

    def to(self, device):
        for v in self.__dict__.values():
            if isinstance(v, torch.nn.Module):
                v.to(device)
            elif isinstance(v, torch.optim.Optimizer):
                for st in v.state.values():
                    for kk, t in st.items():
                        if isinstance(t, torch.Tensor):
                            st[kk] = t.to(device)
        self.device = torch.device(device)
        return self

    def train(self, mode=True):
        for m in self._modules().values():
            m.train(mode)
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def get_saveable(self):
        return self._modules()

    def save(self, ckpt, objects=None):
        saveable = self.get_saveable()
        names = saveable.keys() if objects is None else objects
        out = {k: saveable[k].state_dict() for k in names}
        if hasattr(self, "cfg"):
            out["config"] = self.cfg
        ckpt_save(out, ckpt)

    def load(self, ckpt, cfg=None, objects=None):
        data = ckpt_load(ckpt)
        if cfg is None:
            cfg = data.pop("config")
        self.build(cfg)
        saveable = self.get_saveable()
        names = saveable.keys() if objects is None else objects
        for k in names:
            saveable[k].load_state_dict(data[k])
