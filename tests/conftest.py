import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def as_t(a, complex_last=False):
    t = torch.from_numpy(np.asarray(a))
    if complex_last:
        t = torch.view_as_complex(t.contiguous())
    return t


def philox(name, shape, seed=0, lo=-1.0, hi=1.0):
    from spatialalignmentnetwork_amd import synth
    g = synth._rng(name, seed)
    return torch.from_numpy(g.uniform(lo, hi, shape)).float()


def cplx(name, shape, seed=0):
    return torch.complex(philox(name + ".re", shape, seed), philox(name + ".im", shape, seed))


def rel_err(a, b):
    a = a.double() if not torch.is_complex(a) else a.to(torch.complex128)
    b = b.double() if not torch.is_complex(b) else b.to(torch.complex128)
    den = b.abs().pow(2).sum().sqrt().item()
    return ((a - b).abs().pow(2).sum().sqrt().item()) / max(den, 1e-30)


@pytest.fixture(scope="session")
def ops_golden():
    return load_golden("ops_small.npz")
