"""Round-4 CPU checks of the oracle against reference-made fixtures: metric_SSIM pinned to skimage's algorithm."""
import pytest
import torch

from conftest import load_golden
from oracle import cpu_ref as O


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_ssim_matches_the_skimage_algorithm(tag):
    """The reference's metric (metrics.py:40-43) is skimage.metrics.structural_similarity per slice, averaged.  The fixture holds
    skimage's algorithm restated on the scipy filter it is built from (tests/golden/make_golden.py); 1 - ssimloss (ssimloss.py:
    the same 7 x 7 window, constants and covariance normalisation, valid region = skimage's cropped border) agrees to 1e-6."""
    gold = load_golden("metrics.npz")
    gt, pred = torch.from_numpy(gold[f"{tag}.gt"]), torch.from_numpy(gold[f"{tag}.pred"])
    want = float(gold[f"{tag}.ssim_skimage_algorithm"])
    assert abs((1.0 - O.ssimloss(gt, pred).item()) - want) < 1e-6
