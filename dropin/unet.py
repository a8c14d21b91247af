"""Top-level alias of spatialalignmentnetwork_amd.unet: UNet, CatSequential, ResSequential, Conv2d, Up, Down (unet.py:6-24,119-189).

Put this directory FIRST on PYTHONPATH and the reference's unmodified train.py / eval.py import lists
(train.py:19-22, eval.py:10-13: ``from basemodel import Config``, ``from model import CSModel``,
``from augment import augment``) resolve to the MI355X path.  No code lives here."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(1, _root)

from spatialalignmentnetwork_amd import unet as _impl  # noqa: E402
from spatialalignmentnetwork_amd.unet import *  # noqa: E402,F401,F403

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
