"""The drop-in import surface: with dropin/ first on PYTHONPATH, the reference scripts' import lists (train.py:19-22,
eval.py:10-13) resolve to this package.  CPU only (importing loads libsan_hip.so, which needs no GPU)."""
import os
import subprocess
import sys

from conftest import ROOT

CODE = r'''
import sys
from basemodel import Config
from model import CSModel
from augment import augment
import model, basemodel, augment as aug_mod, varnet, cross, unet, signal_utils, ssimloss, lnccloss, masks, metrics
import spatialalignmentnetwork_amd as pkg
assert CSModel is pkg.model.CSModel and Config is pkg.basemodel.Config and augment is pkg.augment.augment
assert varnet.VarNet is pkg.varnet.VarNet and cross.SpatialTransformer is pkg.cross.SpatialTransformer
assert unet.UNet is pkg.unet.UNet and signal_utils.fft2 is pkg.signal_utils.fft2 and ssimloss.ssimloss is pkg.ssimloss.ssimloss
assert lnccloss.lncc_loss is pkg.lnccloss.lncc_loss and masks.masks is pkg.masks.masks and metrics.mi is pkg.metrics.mi
assert model.gradient_loss is pkg.model.gradient_loss and basemodel.ckpt_load is pkg.basemodel.ckpt_load
cfg = Config(sparsity=0.25, lr=1e-4, shape=32, coils=1, reg="Rec", mask="equispaced", weight_smooth=1000.0, weight_gan=0.0,
             weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=1, chans=4, sens_chans=2, pools=2, sens_pools=2)
net = CSModel(cfg)                      # model.py:39-87 constructor protocol
for attr in ("cfg", "net_mask", "net_T", "net_R", "optim_T", "optim_R", "use_amp", "training"):
    assert hasattr(net, attr), attr
for meth in ("to", "train", "eval", "set_input", "update", "test", "get_vis", "save", "load"):
    assert callable(getattr(net, meth)), meth
try:
    CSModel(Config(**{**{k: cfg[k] for k in cfg.memo}, "mask": "taylor"}))
    raise SystemExit("taylor mask should be refused")
except NotImplementedError:
    pass
print("DROPIN-OK")
'''


def test_reference_import_lists_resolve():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.join(ROOT, "dropin")          # ONLY dropin/: it must find the package itself
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and "DROPIN-OK" in r.stdout, r.stdout + r.stderr
