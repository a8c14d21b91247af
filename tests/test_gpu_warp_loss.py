"""GPU parity tests (through the C ABI), component: warp / grid_sample, SSIM / LNCC / smoothness losses and their backward, augmentation, metrics (rows a12-a15, f3, f4).
Every test carries the round it was written in as a docstring tag; tolerances are written next to the comparisons."""
import numpy as np
import pytest
import torch
import os
import socket
import warnings
import torch.nn.functional as F
from conftest import as_t, cplx, philox, rel_err, load_golden  # noqa: F401
from gpu_common import (S, g, DEV, _build_nets, _run_pipeline, fp32_convs, _conv_bf16x3_checks, _wgrad_bf16x3_checks, _shapes, _load, probe_idx, _digest_errors_r2, _multicoil_nets, _free_port, _dp_cfg, _dp_worker, _psnr, _e4m3, _w_scale, _fill, _pair, _rec_model, _grads, _dp_worker3, _probe_idx, _digest_errors_r3, _model_r4, _state, _conv_ref64, _rccl_single_worker, _model_r5, _act64, _merge_stats)  # noqa: F401

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------- warp and losses
def test_warp_and_losses_golden(S, ops_golden):
    """[round 1]"""
    img = philox("warp.img", (2, 2, 24, 40), lo=0.0, hi=1.0)
    off = philox("warp.off", (2, 24, 40, 2)) * 0.3
    off[0, :2] += 1.5
    off_nchw = off.permute(0, 3, 1, 2).contiguous()
    out, grid = S.ops.warp(g(img), g(off_nchw))
    assert rel_err(out.cpu(), as_t(ops_golden["warp"])) < 1e-5
    assert torch.allclose(grid.cpu(), as_t(ops_golden["identity_grid"]) + off, atol=3e-7)
    out2 = S.ops.grid_sample(g(img), grid)
    assert torch.equal(out2, out)
    # reflection padding (augmentation path) against ATen
    want = torch.nn.functional.grid_sample(img, grid.cpu(), padding_mode="reflection", align_corners=False)
    assert rel_err(S.ops.grid_sample(g(img), grid, padding="reflection").cpu(), want) < 1e-5
    a = philox("loss.a", (2, 1, 40, 56), lo=0.0, hi=1.0)
    b = (a + 0.1 * philox("loss.b", (2, 1, 40, 56))).clamp(0, 1)
    assert abs(S.ssim.ssimloss(g(a), g(b)).item() - float(ops_golden["ssimloss"])) < 2e-6
    assert abs(S.lncc.lncc_loss(g(a), g(b)).item() - float(ops_golden["lncc"])) < 2e-6
    assert abs(S.lncc.ms_lncc_loss(g(a), g(b)).item() - float(ops_golden["ms_lncc"])) < 2e-6
    gl = S.ops.gradient_loss_nchw(g(off_nchw)).item()
    assert abs(gl - float(ops_golden["gradient_loss"])) < 1e-6 * max(1.0, float(ops_golden["gradient_loss"]))


def test_losses_full_size_properties(S):
    """[round 1] SSIM(x, x) == 1 -> loss 0; LNCC is symmetric; at N=8, 320x320."""
    x = philox("fs.x", (8, 1, 320, 320), lo=0.0, hi=1.0)
    y = philox("fs.y", (8, 1, 320, 320), lo=0.0, hi=1.0)
    assert abs(S.ssim.ssimloss(g(x), g(x)).item()) < 1e-6
    assert abs(S.ssim.ssimloss(g(x), g(y)).item() - S.O.ssimloss(x, y).item()) < 2e-6
    l1, l2 = S.lncc.lncc_loss(g(x), g(y)).item(), S.lncc.lncc_loss(g(y), g(x)).item()
    assert abs(l1 - l2) < 1e-7
    assert abs(l1 - S.O.lncc_loss(x, y).item()) < 2e-6


def test_ssim_backward(S):
    """[round 1]"""
    a = philox("sb.a", (2, 1, 40, 56), lo=0.0, hi=1.0)
    b = (a + 0.1 * philox("sb.b", (2, 1, 40, 56))).clamp(0, 1)
    b64 = b.double().requires_grad_(True)
    (S.O.ssimloss(a.double(), b64) * 0.7).backward()
    got = S.ops.ssim_loss_bwd(g(a), g(b), 0.7)
    assert rel_err(got.cpu(), b64.grad.float()) < 1e-4


def test_warp_and_smoothness_backward(S):
    """[round 1]"""
    img = philox("wb.img", (2, 3, 24, 40), lo=0.0, hi=1.0)
    off = (philox("wb.off", (2, 24, 40, 2)) * 0.2)
    off[0, :2] += 1.5
    gout = philox("wb.g", (2, 3, 24, 40))
    o64 = off.double().requires_grad_(True)
    grid64 = S.O.identity_grid(24, 40, torch.float64) + o64
    out = torch.nn.functional.grid_sample(img.double(), grid64, align_corners=False)
    (out * gout.double()).sum().backward()
    off_nchw = off.permute(0, 3, 1, 2).contiguous()
    _, grid = S.ops.warp(g(img), g(off_nchw))
    got = S.ops.warp_bwd_grid(g(img), grid, g(gout))
    want = o64.grad.permute(0, 3, 1, 2).float()
    assert rel_err(got.cpu(), want) < 2e-4
    # smoothness term, accumulated on top
    o64 = off.double().requires_grad_(True)
    (S.O.gradient_loss(o64) * 1000.0).backward()
    S.ops.gradient_loss_bwd(g(off_nchw), got, 1000.0, True)
    assert rel_err(got.cpu(), want + o64.grad.permute(0, 3, 1, 2).float()) < 2e-4


@pytest.mark.parametrize("tag,mode", [("c24x40", "rigid"), ("c24x40", "bspline"), ("r33x20", "rigid"), ("r33x20", "bspline")])
def test_augment_grid_and_sampling_vs_reference(S, tag, mode):
    """[round 1] san_augment_grid + the reflection samplers against the reference's augment() outputs for the same random
    draws (tests/golden/augment.npz) and against the oracle.  2e-6 abs on the grid, 5e-5 / 1e-5 abs on samples."""
    gold = load_golden("augment.npz")
    shp = {"c24x40": (2, 1, 24, 40), "r33x20": (3, 2, 33, 20)}[tag]
    img = cplx(f"aug.{tag}", shp) if tag.startswith("c") else philox(f"aug.{tag}", shp)
    from spatialalignmentnetwork_amd import augment as A
    aff = A.rigid_affine(gold[f"{tag}.{mode}.r_s"], gold[f"{tag}.{mode}.t_s"], DEV)
    ctrl = g(torch.from_numpy(gold[f"{tag}.{mode}.ctrl"])) if mode == "bspline" else None
    grid = S.ops.augment_grid(aff, ctrl, shp[2], shp[3])
    ref_grid = torch.from_numpy(gold[f"{tag}.{mode}.grid"])
    ref_out = torch.from_numpy(gold[f"{tag}.{mode}.out"])
    assert (grid.cpu() - ref_grid).abs().max() < 2e-6
    out = A.sample(g(img), grid)
    got = torch.view_as_real(out.cpu()) if torch.is_complex(out) else out.cpu()
    assert (got - ref_out).abs().max() < 5e-5
    out2, grid2 = A.augment(g(img), rigid=False, bspline=False, grid=g(ref_grid))
    got2 = torch.view_as_real(out2.cpu()) if torch.is_complex(out2) else out2.cpu()
    assert (got2 - ref_out).abs().max() < 1e-5
    # oracle on the same grid
    o_out, _ = S.O.augment(img, grid=ref_grid)
    o = torch.view_as_real(o_out) if torch.is_complex(o_out) else o_out
    assert (got2 - o).abs().max() < 1e-5


def test_augment_random_draws_full_size(S):
    """[round 1] Full-size property checks: a zero-motion augment is the identity; the drawn grid stays within the
    reference's ranges (|grid - identity| <= translation + rotation*sqrt(2) + 1.35/50 B-spline overshoot)."""
    from spatialalignmentnetwork_amd import augment as A
    n, h, w = 8, 320, 320
    img = g(cplx("aug.full", (n, 1, h, w)))
    ident = S.ops.augment_grid(A.rigid_affine([0.0] * n, [0.0] * n, DEV), None, h, w)
    same = A.sample(img, ident)
    # identity grid: ix = j up to fp32 rounding of ((2j+1)/W - 1 + 1) * W (about 2e-5 pixel at j ~ 300), times the
    # neighbour difference of a uniform(-1, 1) image (<= 2)
    assert (torch.view_as_real(same) - torch.view_as_real(img)).abs().max() < 1e-4
    out, grid = A.augment(img)
    assert out.shape == img.shape and out.dtype == img.dtype and torch.isfinite(torch.view_as_real(out)).all()
    bound = A.TRANSLATION + A.ROTATION * 2 ** 0.5 * 1.01 + 1.35 / A.BSPLINE_SCALE
    assert (grid - ident).abs().max().item() <= bound


@pytest.mark.parametrize("tag", ["a", "b"])
def test_metrics_vs_reference(S, tag):
    """[round 1] mse / mae / nmse / mi on the GPU against the reference's metrics.py numbers; psnr by its definition;
    ssim = 1 - ssimloss against the oracle.  1e-6 relative (double sums on device, float32 inputs)."""
    from spatialalignmentnetwork_amd import metrics as M
    gold = load_golden("metrics.npz")
    gt, pred = torch.from_numpy(gold[f"{tag}.gt"]), torch.from_numpy(gold[f"{tag}.pred"])
    for name, fn in (("mse", M.mse), ("mae", M.mae), ("nmse", M.nmse), ("mi", M.mi)):
        ref = float(gold[f"{tag}.{name}"])
        assert abs(fn(g(gt), g(pred)) - ref) <= 1e-6 * max(1.0, abs(ref)), name
    assert abs(M.psnr(g(gt), g(pred)) - 10 * np.log10(1.0 / float(gold[f"{tag}.mse"]))) < 1e-5
    assert abs(M.ssim(g(gt), g(pred)) - (1.0 - S.O.ssimloss(gt, pred).item())) < 2e-5


# ------------------------------------------------------------------------------------------- losses: backward kernels
def test_loss_backward_vs_reference_autograd(S):
    """[round 3] lncc_loss / ms_lncc_loss / ssimloss .backward() (san_lncc_loss_bwd, san_smooth_pool_bwd, san_ssim_loss_bwd_dev) against
    the gradients the REFERENCE's autograd produced (lnccloss.py:7-65, ssimloss.py:11-40), both arguments, with a
    non-trivial upstream gradient kept on the device.  Bar 2e-5."""
    gold = load_golden("autograd_ops.npz")
    for name, fn in (("lncc", S.lncc.lncc_loss), ("ms_lncc", S.lncc.ms_lncc_loss), ("ssim", S.ssim.ssimloss)):
        a, b = _pair()
        a, b = g(a).requires_grad_(True), g(b).requires_grad_(True)
        loss = fn(a, b)
        assert loss.grad_fn is not None
        (loss * 1.7).backward()
        assert abs(loss.item() - float(gold[f"{name}.loss"])) < 2e-6
        ea, eb = rel_err(a.grad.cpu(), as_t(gold[f"{name}.ga"])), rel_err(b.grad.cpu(), as_t(gold[f"{name}.gb"]))
        # the reference's float64 run arbitrates where fp32 itself is ill-conditioned: the coarse scales of ms_lncc divide by
        # near-zero window variances of smoothed images, the reference's own fp32 gradient is 1.0-1.2e-4 from its fp64 one
        ra, rb = rel_err(as_t(gold[f"{name}.ga"]), as_t(gold[f"{name}.ga64"])), rel_err(as_t(gold[f"{name}.gb"]), as_t(gold[f"{name}.gb64"]))
        ea64, eb64 = rel_err(a.grad.cpu(), as_t(gold[f"{name}.ga64"])), rel_err(b.grad.cpu(), as_t(gold[f"{name}.gb64"]))
        print(name, "gradient rel-L2 vs reference fp32", ea, eb, "vs fp64", ea64, eb64, "(reference fp32 vs fp64:", ra, rb, ")")
        # measured: lncc 2.9e-6, ssim 1e-6 vs fp32; ms_lncc 7e-5 vs fp32 with the reference's own fp32-fp64 distance at 1.1e-4
        assert ea < max(2e-5, ra) and eb < max(2e-5, rb), (name, ea, eb)
        assert ea64 < max(2e-5, 1.5 * ra) and eb64 < max(2e-5, 1.5 * rb), (name, ea64, eb64)
    # ragged tiles (37 x 70), uncorrelated pair
    a = g(philox("lncc.a1", (1, 1, 37, 70), lo=0.0, hi=1.0)).requires_grad_(True)
    b = g(philox("lncc.b1", (1, 1, 37, 70), lo=0.0, hi=1.0)).requires_grad_(True)
    S.lncc.lncc_loss(a, b).backward()
    assert rel_err(a.grad.cpu(), as_t(gold["lncc_odd.ga"])) < 2e-5 and rel_err(b.grad.cpu(), as_t(gold["lncc_odd.gb"])) < 2e-5
    # only one side requires a gradient
    a2, b2 = g(philox("lncc.a1", (1, 1, 37, 70), lo=0.0, hi=1.0)), g(philox("lncc.b1", (1, 1, 37, 70), lo=0.0, hi=1.0)).requires_grad_(True)
    S.lncc.lncc_loss(a2, b2).backward()
    assert torch.equal(b2.grad, b.grad) and a2.grad is None


def test_lncc_backward_full_size_vs_oracle_autograd(S):
    """[round 3] The bench batch (N = 8, 320 x 320): LNCC gradients vs oracle autograd (float64), plus symmetry of the kernel pair."""
    x = philox("fs.x", (8, 1, 320, 320), lo=0.0, hi=1.0)
    y = (0.6 * x + 0.4 * philox("fs.y", (8, 1, 320, 320), lo=0.0, hi=1.0))
    x64, y64 = x.double().requires_grad_(True), y.double().requires_grad_(True)
    S.O.lncc_loss(x64, y64).backward()
    gi, gj = S.ops.lncc_loss_bwd(g(x), g(y))
    assert rel_err(gi.cpu(), x64.grad.float()) < 2e-5 and rel_err(gj.cpu(), y64.grad.float()) < 2e-5
    gj2, gi2 = S.ops.lncc_loss_bwd(g(y), g(x))                  # symmetric up to the rounding of the two variance formulas
    assert rel_err(gi2, gi) < 1e-5 and rel_err(gj2, gj) < 1e-5
    # accumulate form
    acc = torch.ones_like(gi)
    S.ops.lncc_loss_bwd(g(x), g(y), gi=acc, gj=None, want_j=False)
    assert torch.allclose(acc, gi + 1.0, rtol=0, atol=1e-6)


def test_lncc_through_warp_and_fft_autograd_vs_reference(S):
    """[round 3] lncc_loss(fixed, net_T.warp(moving, grid)).backward() -> d / d offset incl. the smoothness term, and
    d / d moving (the scatter with float atomics); fft2 / ifft2 / rss adjoints.  Against the reference's autograd."""
    gold = load_golden("autograd_ops.npz")
    st = S.cross.SpatialTransformer(1).to(DEV)
    moving = g(philox("lw.moving", (2, 1, 40, 56), lo=0.0, hi=1.0)).requires_grad_(True)
    fixed = g(philox("lw.fixed", (2, 1, 40, 56), lo=0.0, hi=1.0))
    off = g(philox("lw.off", (2, 40, 56, 2)) * 0.08).requires_grad_(True)
    ident = g(S.O.identity_grid(40, 56))
    loss = S.lncc.lncc_loss(fixed, st.warp(moving, ident + off)) + 3.0 * S.model.gradient_loss(off)
    loss.backward()
    assert abs(loss.item() - float(gold["lw.loss"])) < 2e-6
    e_off, e_mov = rel_err(off.grad.cpu(), as_t(gold["lw.g_off"])), rel_err(moving.grad.cpu(), as_t(gold["lw.g_moving"]))
    print("through-warp gradient errors", e_off, e_mov)
    assert e_off < 2e-5 and e_mov < 2e-5
    x = g(cplx("ag.x", (2, 3, 24, 40))).requires_grad_(True)
    wgt, m = g(philox("ag.w", (2, 1, 24, 40))), g(philox("ag.m", (1, 1, 1, 40)))
    loss = (S.sig.rss(S.sig.ifft2(S.sig.fft2(x) * m)) * wgt).sum()
    loss.backward()
    assert abs(loss.item() - float(gold["fft.loss"])) < 1e-4 * abs(float(gold["fft.loss"]))
    assert rel_err(x.grad.cpu(), as_t(gold["fft.gx"], True)) < 1e-5


@pytest.mark.parametrize("tag", ["a", "b"])
def test_metric_ssim_matches_the_skimage_algorithm(S, tag):
    """[round 4] metrics.py:40-43 averages skimage.metrics.structural_similarity(g[0], p[0], data_range=1) over the batch.  The fixture
    value is skimage's algorithm restated on scipy.ndimage.uniform_filter (make_golden.py::_ssim_skimage_algorithm; skimage is
    not in the image): metric_SSIM (= 1 - ssimloss on the device) must agree to 2e-5 (measured 3e-7)."""
    from conftest import load_golden
    from spatialalignmentnetwork_amd import metrics as M
    gold = load_golden("metrics.npz")
    gt, pred = torch.from_numpy(gold[f"{tag}.gt"]), torch.from_numpy(gold[f"{tag}.pred"])
    want = float(gold[f"{tag}.ssim_skimage_algorithm"])
    got = M.ssim(g(gt), g(pred))
    assert abs(got - want) < 2e-5, (got, want)


def test_test_metrics_single_sync_matches_the_separate_calls(S):
    """[round 4] metrics.test_metrics (what CSModel.test() uses: ONE host synchronisation for MSE / MAE / PSNR / SSIM / MI) against the
    per-metric functions (one synchronisation each): identical values."""
    from spatialalignmentnetwork_amd import metrics as M
    gt = g(philox("tm.gt", (3, 1, 48, 64), lo=0.0, hi=1.0))
    pred = (gt + 0.05 * g(philox("tm.d", (3, 1, 48, 64)))).clamp(0, 1)
    warped = (gt + 0.2 * g(philox("tm.w", (3, 1, 48, 64)))).clamp(0, 1)
    m = M.test_metrics(gt, pred, warped)
    assert m["MSE"] == M.mse(gt, pred) and m["MAE"] == M.mae(gt, pred) and m["PSNR"] == M.psnr(gt, pred)
    assert m["MI"] == M.mi(gt, warped) and abs(m["SSIM"] - M.ssim(gt, pred)) < 1e-7


# ------------------------------------------------------------------------------------------------------------- warp(interp=True)
@pytest.mark.parametrize("hg,wg", [(24, 40), (48, 80), (96, 50)])
def test_warp_interp_resizes_like_the_reference(S, hg, wg):
    """[round 5] cross.py:32-38: ``grid_sample`` on a grid of another size, then ``F.interpolate(size=img.shape[2:])`` (nearest) when
    ``interp`` is set.  Against the same two ATen calls on the CPU."""
    n, c, h, w = 2, 3, 48, 80
    gen = torch.Generator().manual_seed(5)
    img = torch.randn(n, c, h, w, generator=gen)
    grid = torch.rand(n, hg, wg, 2, generator=gen) * 2.2 - 1.1
    st = S.cross.SpatialTransformer(channels=c).to(DEV)
    got = st.warp(g(img), g(grid), interp=True)
    want = F.grid_sample(img, grid, align_corners=False)
    if want.shape != img.shape:
        want = F.interpolate(want, size=img.shape[2:])
    assert got.shape == img.shape
    # (the sampler itself is held to 1e-5 against ATen in test_warp_and_losses_golden: weights formed in another order)
    assert (got.cpu() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
    plain = st.warp(g(img), g(grid))                    # without interp: the grid's size, as in the reference
    assert tuple(plain.shape) == (n, c, hg, wg)


# ---------------------------------------------------------------------------------------------- sampler gradient wrt the image
def test_grid_sample_image_gradient_is_bit_reproducible_and_matches_autograd(S):
    """[round 5] d/d img of the bilinear sampler: 64-bit fixed-point integer atomics instead of float atomics -- 20 launches give identical
    bits (the float form differed from launch to launch), and the values equal ATen's autograd to fp32 rounding."""
    n, c, h, w, ho, wo = 3, 2, 96, 80, 64, 112
    gen = torch.Generator().manual_seed(9)
    img = torch.randn(n, c, h, w, generator=gen)
    grid = torch.rand(n, ho, wo, 2, generator=gen) * 2.4 - 1.2
    gout = torch.randn(n, c, ho, wo, generator=gen) * 3.0
    imr = img.clone().requires_grad_(True)
    F.grid_sample(imr, grid, align_corners=False).backward(gout)
    want = imr.grad
    first = S.ops.grid_sample_bwd_img(g(grid), g(gout), (n, c, h, w))
    for _ in range(20):
        again = S.ops.grid_sample_bwd_img(g(grid), g(gout), (n, c, h, w))
        assert torch.equal(first, again)
    err = (first.cpu() - want).abs().max().item()
    # measured 2.6e-6 of the largest value: the bilinear weights come from fp32 coordinate arithmetic in another order than ATen's
    # (the forward sampler is held to 1e-5 the same way); the fixed-point sum itself resolves 2^-40 of max |g|
    assert err < 1e-5 * want.abs().max().item(), err
    zero = S.ops.grid_sample_bwd_img(g(grid), torch.zeros_like(g(gout)), (n, c, h, w))
    assert float(zero.abs().max()) == 0.0
    # through the module-level autograd function, as a caller of SpatialTransformer.warp would get it
    im2 = g(img).requires_grad_(True)
    S.autograd.warp(im2, g(grid)).backward(g(gout))
    assert torch.equal(im2.grad, first)
