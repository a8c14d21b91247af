"""The reference's OWN training and evaluation scripts, executed against this package (CPU container only; VERDICT r5 "missing" item 3).

``/root/reference/train.py`` is loaded as it lies there (it does not travel and is not copied: the test skips wherever the reference
tree is absent, i.e. on the GPU box) with ``dropin/`` in front of it on ``sys.path`` -- so its ``from model import CSModel``, ``from
basemodel import Config``, ``from augment import augment`` resolve to THIS package -- and with stand-ins for what the image lacks and
the path does not need: ``torch.utils.tensorboard``, ``torchvision``, ``paired_dataset`` (synthetic complex64 volumes in the
reference's item format: one array [coils, H, W] per protocol).  ``main(args)`` then runs unmodified: configuration from its argument
names, ``CSModel(cfg=cfg)`` / ``CSModel(ckpt=..., cfg=cfg, objects=...)``, ``net.to(device)``, the epoch loop with ``net.train()``,
``net.set_input(*batch)``, ``net.update()``, ``get_vis('scalars' / 'histograms' / 'images')``, ``net.eval()``, ``net.test()``,
``net.save(...)`` and the resume path.

There is no GPU here and the hot path has no CPU fallback, so the three methods that launch kernels (set_input / update / test) are
replaced by recorders on a SUBCLASS of the real CSModel; everything else -- constructor, Config, optimisers, ``to``, ``train`` / ``eval``,
``get_vis``, ``save``, ``load`` -- is the product code.  What this pins is the ARGUMENT FLOW of train.py:63-332 into the class
surface (tensor ranks, dtypes, call order, checkpoint round trip); the arithmetic behind the three recorders is what the GPU
suite tests (tests/test_gpu_step_runtime.py runs the same loop shape on the real methods)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"

HARNESS = r'''
import importlib.util, json, os, sys, types
sys.dont_write_bytecode = True
import numpy as np
import torch
ROOT, REF, LOGDIR, RESUME = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
sys.path[:0] = [os.path.join(ROOT, "dropin"), REF]

calls = []                                            # (what, details) in call order

# ---- stand-ins for modules the image lacks (none of them is on the hot path)
tb = types.ModuleType("torch.utils.tensorboard")
class SummaryWriter:
    def __init__(self, logdir): calls.append(("writer", logdir))
    def add_text(self, *a, **k): pass
    def add_scalar(self, tag, val, step): calls.append(("scalar", tag, float(val), int(step)))
    def add_histogram(self, tag=None, global_step=None, **k): calls.append(("histogram", tag, sorted(k)))
    def flush(self): pass
    def close(self): pass
tb.SummaryWriter = SummaryWriter
sys.modules["torch.utils.tensorboard"] = tb
torch.utils.tensorboard = tb
tv, tvu = types.ModuleType("torchvision"), types.ModuleType("torchvision.utils")
def save_image(val, path, **k): calls.append(("save_image", os.path.basename(path).split("_", 1)[1], tuple(val.shape), sorted(k)))
tvu.save_image = save_image
tv.utils = tvu
sys.modules["torchvision"], sys.modules["torchvision.utils"] = tv, tvu
pd = types.ModuleType("paired_dataset")
class _Volume(torch.utils.data.Dataset):
    """Items in the reference's format (paired_dataset.py:101-103): a list with one complex64 array [coils, crop, crop] per protocol."""
    def __init__(self, n, coils, crop, seed):
        g = np.random.default_rng(seed)
        self.items = [[(g.random((coils, crop, crop)) + 0j).astype(np.complex64) for _ in range(2)] for _ in range(n)]
    def __len__(self): return len(self.items)
    def __getitem__(self, i): return self.items[i]
def get_paired_volume_datasets(path, crop=None, protocals=None, **k):
    calls.append(("dataset", path, int(crop), protocals))
    return [_Volume(6, 1, int(crop), 7), _Volume(6, 1, int(crop), 8)]
def center_crop(x, size):
    h, w = x.shape[-2:]
    th, tw = size
    i, j = (h - th) // 2, (w - tw) // 2
    return x[..., i:i + th, j:j + tw]
pd.get_paired_volume_datasets, pd.center_crop = get_paired_volume_datasets, center_crop
sys.modules["paired_dataset"] = pd

nib = types.ModuleType("nibabel")
class Nifti1Image:
    def __init__(self, data, affine): self.shape, self.affine = tuple(data.shape), affine
def nib_save(img, path): calls.append(("nifti", os.path.basename(path), img.shape))
nib.Nifti1Image, nib.save = Nifti1Image, nib_save
sys.modules["nibabel"] = nib

# ---- the reference script, as it lies there
SCRIPT = sys.argv[5] if len(sys.argv) > 5 else "train.py"
spec = importlib.util.spec_from_file_location("ref_train", os.path.join(REF, SCRIPT))
train = importlib.util.module_from_spec(spec)
spec.loader.exec_module(train)
import spatialalignmentnetwork_amd as pkg
assert train.CSModel is pkg.model.CSModel and train.Config is pkg.basemodel.Config and train.augment is pkg.augment.augment

# ---- `torch.device('cuda')` inside main() -> the CPU of this container (only there: the package keeps the real torch)
class _TorchProxy:
    def __getattr__(self, name): return getattr(torch, name)
    @staticmethod
    def device(name): return torch.device("cpu" if str(name).startswith("cuda") else name)
train.torch = _TorchProxy()

# ---- the three kernel-launching methods become recorders; everything else is the product class
class Recorder(pkg.model.CSModel):
    def set_input(self, img_full, img_aux=None):
        for name in [k for k in self.__dict__ if k.startswith(("loss_", "img_", "metric_"))]:
            delattr(self, name)
        calls.append(("set_input", tuple(img_full.shape), str(img_full.dtype), None if img_aux is None else tuple(img_aux.shape),
                      self.training, str(img_full.device)))
        assert torch.is_complex(img_full) and img_full.shape[1] == self.cfg.coils and img_full.shape[-1] == self.cfg.shape
        n, _, h, w = img_full.shape
        self.img_full, self.img_aux = img_full, img_aux
        for k in ("img_full_rss", "img_sampled_rss", "img_aux_rss", "img_warped_rss", "img_rec"):
            setattr(self, k, torch.full((n, 1, h, w), 0.5))
        self.img_offset = torch.zeros((n, h, w, 2))
    def update(self):
        calls.append(("update", self.training, self.cfg.reg))
        assert self.training is True
        self.loss_sim, self.loss_smooth = torch.tensor(0.25), torch.tensor(0.125)
    def test(self):
        calls.append(("test", self.training))
        assert self.training is False
        self.loss_sim, self.loss_all = torch.tensor(0.25), torch.tensor(0.25)
        self.metric_PSNR, self.metric_SSIM, self.metric_MI = 30.0, 0.9, 1.0
        return -30.0
    def save(self, ckpt, objects=None):
        calls.append(("save", os.path.basename(ckpt)))
        return super().save(ckpt, objects)
train.CSModel = Recorder

args = types.SimpleNamespace(logdir=LOGDIR, resume=(None if RESUME == "-" else RESUME), load_nets=(None if RESUME == "-" else ["net_R", "net_T"]),
                             epoch=1, batch_size=2, num_workers=0, lr=1e-4, intel_stop=1, reg="Rec", smooth_weight=1000.0,
                             gan_weight=0.0, gan_sim_weight=0.0, sim_weight=1.0, mask="equispaced", sparsity=0.25, train="/data/train",
                             val="/data/val", crop=32, coils=1, protocals=["T2", "T1"], aux_aug="None", prefetch=True, use_amp=False,
                             force_gpu=True)
if SCRIPT == "eval.py":
    os.makedirs(os.path.join(LOGDIR, "nii"), exist_ok=True)
    args = types.SimpleNamespace(resume=RESUME, val="/data/val", protocals=["T2", "T1"], aux_aug=0, save=os.path.join(LOGDIR, "nii"),
                                 metric=os.path.join(LOGDIR, "metric.json"))
# (train.py, one pass: 12 training slices in batches of 2 -> 6 iterations; validation; intel_stop writes best.pt; final checkpoint)
train.main(args)
print("CALLS " + json.dumps(calls))
'''


def _run(tmp_path, resume="-", script="train.py"):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    os.makedirs(str(tmp_path), exist_ok=True)
    r = subprocess.run([sys.executable, "-c", HARNESS, ROOT, REF, str(tmp_path), resume, script], env=env, capture_output=True, text=True,
                       cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("CALLS ")][-1]
    return json.loads(line[6:])


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train.py")), reason="the reference tree is only present in the build container")
def test_reference_train_main_drives_the_dropin_model(tmp_path):
    calls = _run(tmp_path)
    kinds = [c[0] for c in calls]
    # data: training volumes are cropped 10 % larger than the model's shape (train.py:142-143), validation exactly
    assert [c[2] for c in calls if c[0] == "dataset"] == [35, 32]
    # six training iterations of batch 2: train() -> set_input(full, aux) complex64 [2, 1, 32, 32] on the (stand-in) device -> update()
    tr = [c for c in calls if c[0] == "set_input" and c[4] is True]
    assert len(tr) == 6 and all(c[1] == [2, 1, 32, 32] and c[2] == "torch.complex64" and c[3] == [2, 1, 32, 32] for c in tr)
    assert kinds.count("update") == 6 and all(c[1] is True and c[2] == "Rec" for c in calls if c[0] == "update")
    for i, k in enumerate(kinds):
        if k == "update":
            assert kinds[i - 1] == "set_input"
    # validation: eval() -> set_input -> test() for the six validation batches, scalars through get_vis into the writer
    assert kinds.count("test") == 6 and all(c[1] is False for c in calls if c[0] == "test")
    val = [c for c in calls if c[0] == "scalar" and c[1].startswith("val/")]
    assert {c[1] for c in val} >= {"val/loss_sim", "val/metric_PSNR", "val/metric_SSIM", "val/metric_MI"}
    assert all(c[3] == 6 for c in val)
    # checkpoints: best.pt (intel_stop) and the final one, written by the product's save() as a directory of npz blobs + config
    assert [c[1] for c in calls if c[0] == "save"] == ["best.pt", "ckpt_0000000006.pt"]
    ck = os.path.join(str(tmp_path), "ckpt", "ckpt_0000000006.pt")
    assert os.path.isdir(ck) and {"config", "net_R", "net_T", "net_mask"} <= set(os.listdir(ck))
    cfg = json.load(open(os.path.join(ck, "config")))
    assert cfg["shape"] == 32 and cfg["reg"] == "Rec" and cfg["weight_smooth"] == 1000.0 and cfg["sparsity"] == 0.25
    # ... and train.py's own resume path loads it back into a new model (CSModel(ckpt=..., cfg=cfg, objects=[...]))
    # (into a fresh log directory: like the reference's, the product's save() refuses to overwrite an existing checkpoint)
    calls2 = _run(tmp_path / "resumed", resume=ck)
    assert [c[0] for c in calls2].count("update") == 6

    # ... and the reference's eval.py (eval.py:29-80) on the same checkpoint: CSModel(ckpt=...), use_amp off, eval(), one set_input +
    # test() per VOLUME (all its slices as one batch), scalars into the metric file, the six image attributes into NIfTI volumes
    ev = _run(tmp_path / "evaluated", resume=ck, script="eval.py")
    si = [c for c in ev if c[0] == "set_input"]
    assert len(si) == 2 and all(c[1] == [6, 1, 32, 32] and c[2] == "torch.complex64" and c[4] is False for c in si)
    assert [c[0] for c in ev].count("test") == 2 and [c[0] for c in ev].count("update") == 0
    names = [c[1] for c in ev if c[0] == "nifti"]
    assert names == [f"{i}_{k}.nii" for i in range(2) for k in ("image", "aux", "sampled", "warped", "rec", "grid")]
    assert [c[2] for c in ev if c[0] == "nifti" and c[1] == "0_grid.nii"] == [[32, 32, 6, 1, 3]]
    metric = json.load(open(os.path.join(str(tmp_path / "evaluated"), "metric.json")))
    assert len(metric) == 2 and {"loss_sim", "metric_PSNR", "metric_SSIM", "metric_MI"} <= set(metric[0])
