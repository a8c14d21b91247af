"""GPU parity tests (through the C ABI), component: convolution kernels: 3x3 / 1x1 / transposed forward, data and weight gradients, every operand format (rows a9, N1).
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


# ------------------------------------------------------------ conv/norm stack
@pytest.mark.parametrize("cin,cout,h,w,ks", [(3, 18, 32, 32, 3), (18, 18, 64, 64, 3), (36, 18, 40, 40, 3),
                                              (72, 144, 20, 20, 3), (5, 7, 24, 40, 3), (18, 2, 32, 32, 1),
                                              (64, 64, 40, 40, 1), (96, 32, 64, 32, 3), (2, 32, 320, 320, 3)])
def test_conv2d_vs_torch(S, cin, cout, h, w, ks):
    """[round 1]"""
    n = 2
    x = philox("cv.x", (n, cin, h, w))
    wt = philox("cv.w", (cout, cin, ks, ks)) * (1.0 / (cin * ks * ks) ** 0.5)
    b = philox("cv.b", (cout,))
    sc, sh = philox("cv.sc", (n, cin), lo=0.5, hi=1.5), philox("cv.sh", (n, cin))
    xin = S.ops.Act(g(x), 0, cin, g(sc), g(sh), 0.2)
    y = torch.empty((n, cout, h, w), device=DEV)
    part = S.ops.conv2d(xin, g(wt), g(b), S.ops.full(y), stats=True)
    xa = torch.nn.functional.leaky_relu(x * sc[:, :, None, None] + sh[:, :, None, None], 0.2)
    want = torch.nn.functional.conv2d(xa.double(), wt.double(), b.double(), padding=ks // 2).float()
    assert rel_err(y.cpu(), want) < 2e-6
    # fused statistics: merged partials == mean / biased variance of the output
    scale = torch.empty((n, cout), device=DEV)
    shift = torch.empty((n, cout), device=DEV)
    S.ops.norm_finalize(part, S.ops.NORM_INSTANCE, 1e-5, scale, shift, 0)
    mean = want.double().mean(dim=(2, 3))
    var = want.double().var(dim=(2, 3), unbiased=False)
    wsc = 1.0 / torch.sqrt(var + 1e-5)
    assert torch.allclose(scale.cpu().double(), wsc, rtol=2e-5)
    assert torch.allclose(shift.cpu().double(), -mean * wsc, rtol=2e-4, atol=2e-5)


def test_conv_blocks_golden(S, ops_golden):
    """[round 1]"""
    p = S.synth.fill_params([("layers.0.weight", (6, 3, 3, 3)), ("layers.3.weight", (6, 6, 3, 3))], seed=11)
    cb = S.varnet.ConvBlock(3, 6)
    cb.load_state_dict(p)
    cb.to(DEV)
    y = cb(g(philox("cb.x", (2, 3, 24, 40))))
    assert rel_err(y.cpu(), as_t(ops_golden["convblock"])) < 1e-5
    p = S.synth.fill_params([("layers.0.weight", (6, 4, 2, 2))], seed=12)
    tb = S.varnet.TransposeConvBlock(6, 4)
    tb.load_state_dict(p)
    tb.to(DEV)
    y = tb(g(philox("tb.x", (2, 6, 12, 20))))
    assert rel_err(y.cpu(), as_t(ops_golden["tconvblock"])) < 1e-5


@pytest.mark.parametrize("cin,cout,h,w", [(36, 18, 20, 20), (288, 144, 20, 20), (8, 4, 16, 24), (5, 3, 7, 9), (72, 36, 33, 50),
                                          (16, 8, 16, 24), (144, 72, 9, 17)])
def test_tconv_vs_torch(S, cin, cout, h, w):
    """[round 1]"""
    n = 2
    x = philox("tc.x", (n, cin, h, w))
    wt = philox("tc.w", (cin, cout, 2, 2)) * (1.0 / cin ** 0.5)
    y = torch.empty((n, cout, 2 * h, 2 * w), device=DEV)
    part = S.ops.tconv2x2(S.ops.full(g(x)), g(wt), S.ops.full(y), stats=True)
    want = torch.nn.functional.conv_transpose2d(x.double(), wt.double(), stride=2).float()
    assert rel_err(y.cpu(), want) < 2e-6
    scale = torch.empty((n, cout), device=DEV)
    shift = torch.empty((n, cout), device=DEV)
    S.ops.norm_finalize(part, S.ops.NORM_INSTANCE, 1e-5, scale, shift, 0)
    var = want.double().var(dim=(2, 3), unbiased=False)
    assert torch.allclose(scale.cpu().double(), 1.0 / torch.sqrt(var + 1e-5), rtol=2e-5)


# ------------------------------------------------------- backward building blocks
@pytest.mark.parametrize("cin,cout,h,w,ks", [(3, 18, 32, 32, 3), (18, 18, 64, 64, 3), (36, 18, 40, 24, 3),
                                              (20, 9, 17, 70, 3), (18, 2, 32, 32, 1), (64, 64, 20, 20, 1)])
def test_conv_dgrad_wgrad_vs_autograd(S, cin, cout, h, w, ks):
    """[round 1]"""
    n = 2
    x = philox("bw.x", (n, cin, h, w))
    wt = philox("bw.w", (cout, cin, ks, ks)) * (1.0 / (cin * ks * ks) ** 0.5)
    sc, sh = philox("bw.sc", (n, cin), lo=0.5, hi=1.5), philox("bw.sh", (n, cin))
    dy = philox("bw.dy", (n, cout, h, w))
    xa = torch.nn.functional.leaky_relu(x * sc[:, :, None, None] + sh[:, :, None, None], 0.2).double().requires_grad_(True)
    w64 = wt.double().requires_grad_(True)
    torch.nn.functional.conv2d(xa, w64, None, padding=ks // 2).backward(dy.double())
    dx = torch.empty((n, cin, h, w), device=DEV)
    S.ops.conv2d_dgrad(S.ops.full(g(dy)), g(wt), S.ops.full(dx))
    assert rel_err(dx.cpu(), xa.grad.float()) < 3e-6
    dw = torch.empty((cout, cin, ks, ks), device=DEV)
    S.ops.conv2d_wgrad(S.ops.Act(g(x), 0, cin, g(sc), g(sh), 0.2), S.ops.full(g(dy)), dw)
    assert rel_err(dw.cpu(), w64.grad.float()) < 1e-5
    S.ops.conv2d_wgrad(S.ops.Act(g(x), 0, cin, g(sc), g(sh), 0.2), S.ops.full(g(dy)), dw, accumulate=True)
    assert rel_err(dw.cpu(), 2 * w64.grad.float()) < 1e-5


@pytest.mark.parametrize("n,cin,cout,h,w,ks,slope,affine,coff", [
    (3, 72, 72, 20, 20, 3, 0.2, True, 0),      # several input / output channel blocks, balanced 10-row tiles
    (1, 40, 100, 48, 16, 3, 0.0, True, 4),     # ReLU, channel views with an offset, one tile column
    (2, 8, 16, 36, 52, 3, 1.0, False, 0),      # no lazy affine (materialised input), partial tiles in x and y
    (2, 18, 18, 24, 24, 3, -0.5, True, 0),     # slope outside [0, 1]: generic kernel
    (2, 288, 40, 8, 12, 1, 0.01, True, 2),     # 1x1, many input blocks
])
def test_conv_wgrad_paths(S, n, cin, cout, h, w, ks, slope, affine, coff):
    """[round 1] Weight gradient through the pipelined kernel's channel blocking / tile geometry variants and the
    generic fallback, against float64 autograd.  Tolerance 1e-5 relative (fp32 sums of up to 2e4 terms)."""
    x = philox("wp.x", (n, cin + coff, h, w))
    dy = philox("wp.dy", (n, cout + coff, h, w))
    sc, sh = philox("wp.sc", (n, cin + coff), lo=0.5, hi=1.5), philox("wp.sh", (n, cin + coff))
    xs = x[:, coff:]
    if affine:
        xs = xs * sc[:, coff:, None, None] + sh[:, coff:, None, None]
    xa = torch.nn.functional.leaky_relu(xs, slope).double() if slope != 1.0 else xs.double()
    w64 = torch.zeros((cout, cin, ks, ks), dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xa, w64, None, padding=ks // 2).backward(dy[:, coff:].double())
    dw = torch.empty((cout, cin, ks, ks), device=DEV)
    act = S.ops.Act(g(x), coff, cin, g(sc) if affine else None, g(sh) if affine else None, slope)
    S.ops.conv2d_wgrad(act, S.ops.Act(g(dy), coff, cout, None, None, 1.0), dw)
    assert rel_err(dw.cpu(), w64.grad.float()) < 1e-5


@pytest.mark.parametrize("wd", [-1, 0, 1], ids=["auto", "lds-weights", "direct-weights"])
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 24, 48, 16, 32), (1, 72, 72, 20, 44), (2, 36, 72, 40, 40), (1, 96, 32, 33, 50),
                                            (2, 144, 144, 24, 24), (1, 288, 144, 16, 16), (1, 64, 64, 9, 17), (2, 18, 18, 48, 64),
                                            (1, 36, 18, 64, 64), (2, 18, 36, 40, 48)])
def test_conv_bf16x3_vs_float64(S, n, cin, cout, h, w, wd):
    """[round 1] The bf16 matrix-core convolution with three-way split operands (csrc/san_conv_bf16.hip) against float64:
    forward with lazy affine + LeakyReLU input, bias, channel views and fused statistics, and the data gradient.
    Bars: 3e-6 relative on outputs (fp32-level: the split drops O(2^-24) terms), 2e-5 on merged statistics."""
    ops = S.ops
    assert ops.bf16x3_eligible(cin, cout, h, w, 3)
    ops.lib().call("san_conv_bf16x3_set_tuning", wd, -1)
    try:
        _conv_bf16x3_checks(ops, n, cin, cout, h, w)
    finally:
        ops.lib().call("san_conv_bf16x3_set_tuning", -1, -1)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 16, 32), (1, 576, 288, 10, 20), (2, 72, 36, 40, 40), (1, 32, 64, 33, 50),
                                            (2, 20, 50, 9, 17), (1, 144, 72, 24, 40)])
def test_conv1x1_bf16x3_vs_float64(S, n, cin, cout, h, w):
    """[round 1] The bf16x3 kernel as a 1x1 convolution (alignment-net 1x1 layers, data gradient of the transposed convolutions):
    forward with lazy affine + LeakyReLU input through a channel view, bias, fused statistics, and the 1x1 data
    gradient, against float64.  Bars as for the 3x3 form: 3e-6 on outputs, 2e-5 on merged statistics."""
    ops = S.ops
    assert ops.bf16x3_eligible(cin, cout, h, w, 1)
    x = philox("c1.x", (n, cin + 3, h, w))
    wt = philox("c1.w", (cout, cin, 1, 1)) * (1.0 / cin ** 0.5)
    sc, sh = philox("c1.sc", (n, cin + 3), lo=0.5, hi=1.5), philox("c1.sh", (n, cin + 3))
    b = philox("c1.b", (cout,))
    y = torch.empty((n, cout + 2, h, w), device=DEV)
    part = ops.conv2d(ops.Act(g(x), 3, cin, g(sc), g(sh), 0.2), g(wt), g(b), ops.Act(y, 2, cout, None, None, 1.0), stats=True)
    act = torch.nn.functional.leaky_relu(x[:, 3:] * sc[:, 3:, None, None] + sh[:, 3:, None, None], 0.2).double()
    ref = torch.nn.functional.conv2d(act, wt.double(), b.double())
    assert rel_err(y[:, 2:].cpu(), ref.float()) < 3e-6
    p = part.cpu().double()
    cnt, mean_t, m2_t = p[..., 0], p[..., 1], p[..., 2]
    tot = cnt.sum(-1)
    assert torch.all(tot == h * w)
    mean = (cnt * mean_t).sum(-1) / tot
    m2 = (m2_t + cnt * (mean_t - mean[..., None]) ** 2).sum(-1)
    assert (mean - ref.mean(dim=(2, 3))).abs().max() < 2e-5
    assert rel_err((m2 / tot).float(), ref.var(dim=(2, 3), unbiased=False).float()) < 2e-5
    dy = philox("c1.dy", (n, cout, h, w))
    dx = torch.empty((n, cin, h, w), device=DEV)
    ops.conv2d_dgrad(ops.full(g(dy)), g(wt), ops.full(dx))
    ref_dx = torch.einsum("nohw,oi->nihw", dy.double(), wt.double()[:, :, 0, 0])
    assert rel_err(dx.cpu(), ref_dx.float()) < 3e-6


@pytest.mark.parametrize("mode", [1, 0], ids=["direct", "split"])
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 72, 72, 20, 44), (2, 36, 72, 40, 40), (1, 96, 32, 33, 52), (2, 144, 144, 24, 24),
                                            (4, 288, 144, 16, 16), (8, 40, 50, 14, 12), (2, 32, 48, 31, 31), (2, 64, 64, 30, 46),
                                            (2, 3, 18, 32, 40), (3, 18, 2, 17, 36), (1, 8, 8, 64, 64)])
def test_wgrad_bf16x3_vs_float64(S, n, cin, cout, h, w, mode):
    """[round 1] The bf16 matrix-core weight gradient with three-way split operands (csrc/san_wgrad_bf16.hip) against float64:
    lazily activated input read through a channel view, dy through a channel view, widths that are not a multiple
    of 8 or 4, ragged row bands, overwrite and accumulate, in both forms of the kernel (direct: operands split in
    registers; split: bf16 planes written first -- the form taken when W % 4 != 0 whatever the mode).
    Bar: 3e-6 relative L2 and 3e-6 of the largest entry."""
    ops = S.ops
    ops.lib().call("san_conv_wgrad_bf16x3_set_mode", mode)
    x = philox("wb.x", (n, cin + 3, h, w))
    dy = philox("wb.dy", (n, cout + 2, h, w))
    sc, sh = philox("wb.sc", (n, cin + 3), lo=0.5, hi=1.5), philox("wb.sh", (n, cin + 3))
    act = torch.nn.functional.leaky_relu(x[:, 3:] * sc[:, 3:, None, None] + sh[:, 3:, None, None], 0.2).double()
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.double, requires_grad=True)
    torch.nn.functional.conv2d(act, wt, padding=1).backward(dy[:, 2:].double())
    ref = wt.grad
    dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
    xa, da = ops.Act(g(x), 3, cin, g(sc), g(sh), 0.2), ops.Act(g(dy), 2, cout, None, None, 1.0)
    try:
        _wgrad_bf16x3_checks(ops, xa, da, dw, ref)
    finally:
        ops.lib().call("san_conv_wgrad_bf16x3_set_mode", -1)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 288, 576, 10, 12), (2, 36, 72, 40, 40), (3, 64, 64, 17, 20), (1, 32, 64, 33, 52),
                                            (8, 20, 50, 7, 12), (1, 144, 288, 6, 6)])
def test_wgrad1x1_bf16x3_vs_float64(S, n, cin, cout, h, w):
    """[round 1] The 1x1 weight gradient on the bf16 matrix cores (three-way split operands) against float64: channel views,
    lazily activated input, plane sizes that are not a multiple of the 32-pixel step, ragged channel blocks,
    overwrite / accumulate, bit-reproducibility.  Bar: 3e-6 relative L2 and of the largest entry."""
    ops = S.ops
    x = philox("w1.x", (n, cin + 3, h, w))
    dy = philox("w1.dy", (n, cout + 2, h, w))
    sc, sh = philox("w1.sc", (n, cin + 3), lo=0.5, hi=1.5), philox("w1.sh", (n, cin + 3))
    act = torch.nn.functional.leaky_relu(x[:, 3:] * sc[:, 3:, None, None] + sh[:, 3:, None, None], 0.2).double()
    ref = torch.einsum("nohw,nihw->oi", dy[:, 2:].double(), act)[:, :, None, None]
    dw = torch.full((cout, cin, 1, 1), float("nan"), device=DEV)
    xa, da = ops.Act(g(x), 3, cin, g(sc), g(sh), 0.2), ops.Act(g(dy), 2, cout, None, None, 1.0)
    ops.conv2d_wgrad1x1_bf16x3(xa, da, dw)
    got = dw.cpu().double()
    assert ((got - ref).norm() / ref.norm()).item() < 3e-6
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 3e-6
    ops.conv2d_wgrad1x1_bf16x3(xa, da, dw, accumulate=True)
    assert ((dw.cpu().double() - 2 * ref).norm() / ref.norm()).item() < 6e-6
    dw2 = torch.empty_like(dw)
    ops.conv2d_wgrad1x1_bf16x3(xa, da, dw2)
    assert torch.equal(dw2.cpu().double(), got)
    # transposed write (the ConvTranspose2d weight layout [Cin, Cout']), accumulating
    dwt = torch.ones((cin, cout), device=DEV)
    ops.conv2d_wgrad1x1_bf16x3(xa, da, dwt, accumulate=True, transposed=True)
    assert torch.allclose(dwt.cpu().double() - 1.0, got[:, :, 0, 0].t(), rtol=0, atol=1e-6 * float(ref.abs().max()))


@pytest.mark.parametrize("mode,bar_conv,bar_wgrad", [("bf16x2", 3e-5, 3e-5), ("bf16", 6e-3, 6e-3)])
def test_conv_precision_modes_layers(S, mode, bar_conv, bar_wgrad):
    """[round 2] The two- and one-part forms of the matrix-core convolution / weight gradient against float64: two bf16 parts carry
    16 mantissa bits (2^-17 = 7.6e-6 per operand), one part 8 bits (2^-9 = 2e-3).  Measured values are printed."""
    ops = S.ops
    n, cin, cout, h, w = 2, 72, 36, 40, 40
    x, wt = philox("np.x", (n, cin, h, w)), philox("np.w", (cout, cin, 3, 3)) * 0.05
    dy = philox("np.dy", (n, cout, h, w))
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1)
    x64, w64 = x.double(), wt.double().requires_grad_(True)
    (torch.nn.functional.conv2d(x64, w64, padding=1) * dy.double()).sum().backward()
    try:
        with ops.conv_precision(mode):
            y = torch.empty((n, cout, h, w), device=DEV)
            ops.conv2d(ops.full(g(x)), g(wt), None, ops.full(y))
            dw = torch.zeros((cout, cin, 3, 3), device=DEV)
            ops.conv2d_wgrad_bf16x3(ops.full(g(x)), ops.full(g(dy)), dw)       # (conv2d_wgrad may pick the fp32 kernel here)
            torch.cuda.synchronize()
        e1, e2 = rel_err(y.cpu().double(), ref), rel_err(dw.cpu().double(), w64.grad)
        print(f"{mode}: conv rel-L2 {e1:.2e}, weight gradient rel-L2 {e2:.2e}")
        assert e1 < bar_conv and e2 < bar_wgrad
        # back in the default mode the same call is fp32-equivalent again
        ops.conv2d(ops.full(g(x)), g(wt), None, ops.full(y))
        assert rel_err(y.cpu().double(), ref) < 3e-6
    finally:
        ops.set_conv_precision("bf16x3")


# ------------------------------------------------------------------ fp16 two-part operand format (default fp32-equivalent mode)
@pytest.mark.parametrize("scale", [1.0, 3e-7, 2e4])
@pytest.mark.parametrize("n,cin,cout,h,w,ks", [(2, 72, 36, 40, 40, 3), (1, 18, 18, 64, 64, 3), (2, 64, 64, 16, 32, 1), (1, 288, 144, 20, 20, 3)])
def test_f16x2_forward_and_gradients_vs_float64(S, n, cin, cout, h, w, ks, scale):
    """[round 2] The fp16 two-part forms against float64: forward convolution (activations, no scale needed), data gradient and weight
    gradient with dy of magnitude `scale` x [tiny .. 1] (a 1e-6 dynamic range inside the tensor) scaled by the power of two
    its recorded maximum asks for.  22 mantissa bits: bars 3e-6 like the six-product bf16 form (measured ~3e-7)."""
    ops = S.ops
    assert ops.F16_FWD[0] and ops.F16_BWD[0]
    x = philox("f16.x", (n, cin, h, w)) * 2
    wt = philox("f16.w", (cout, cin, ks, ks)) * 0.1
    pad = ks // 2
    g0 = philox("f16.g", (n, cout, h, w))
    rng = torch.exp(philox("f16.r", (n, cout, h, w)) * 7.0)                  # e^-7 .. e^7 spread inside the tensor
    gout = g0 * rng * (scale / rng.max())
    x64, w64 = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, padding=pad)
    (y64 * gout.double()).sum().backward()
    # forward (two fp16 parts picked automatically for non-gradient inputs)
    y = torch.empty((n, cout, h, w), device=DEV)
    ops.conv2d(ops.full(g(x)), g(wt), None, ops.full(y))
    assert ops.lib().query("san_get_conv_precision") == 3
    assert rel_err(y.cpu().double(), y64.detach()) < 3e-6
    # a dy tensor whose maximum was recorded by the activation backward: identity activation (slope 1, no affine) passes g through
    ops.AMAX.reset(DEV)
    dy = ops.Act(torch.empty((n, cout, h, w), device=DEV), 0, cout)
    ops.act_bwd(ops.full(g(gout)), ops.full(g(philox("f16.y", (n, cout, h, w)))), dy, instance_norm=False)
    assert dy.amax is not None and torch.equal(dy.buf.cpu(), gout)
    got_max = ops.amax_value(dy.amax)
    assert got_max == gout.abs().max().item()
    dx = torch.empty((n, cin, h, w), device=DEV)
    ops.conv2d_dgrad(dy, g(wt), ops.full(dx))
    e_d = rel_err(dx.cpu().double(), x64.grad)
    dw = torch.zeros((cout, cin, ks, ks), device=DEV)
    if ks == 3:
        ops.conv2d_wgrad_bf16x3(ops.full(g(x)), dy, dw)
    else:
        ops.conv2d_wgrad1x1_bf16x3(ops.full(g(x)), dy, dw)
    torch.cuda.synchronize()
    e_w = rel_err(dw.cpu().double(), w64.grad)
    print(f"f16x2 {cin}->{cout} k{ks} @{h}x{w} scale {scale:g}: data gradient {e_d:.2e}, weight gradient {e_w:.2e}")
    assert e_d < 3e-6 and e_w < 3e-6


@pytest.mark.parametrize("n,cin,cout,h,w,ks,kind", [
    (2, 72, 36, 40, 40, 3, "conv"),        # FLAT tile, weights direct
    (2, 18, 18, 64, 96, 3, "conv"),        # 32 x 8 tiles, half-padded channel blocks, operand-swapped epilogue
    (1, 96, 160, 24, 40, 3, "conv"),       # five channel blocks
    (2, 288, 288, 20, 20, 3, "conv"),      # split-K
    (2, 64, 32, 32, 48, 1, "conv"),        # 1x1 form
    (2, 72, 36, 16, 24, 1, "tconv"),       # transposed 2x2 s2: pixel-shuffle epilogue
])
def test_fp8_forward_matches_quantised_float64(S, n, cin, cout, h, w, ks, kind):
    """[round 2] The fp8 mode's forward convolutions against float64 arithmetic on the SAME quantised operands: activations x 8 and
    weights x S_w = 2^(7 - floor(log2 max |w|)) rounded to OCP e4m3 (torch.float8_e4m3fn on the CPU), products and sums in
    float64.  This pins the operand layout of v_mfma_f32_16x16x32_fp8_fp8, the hardware conversion (round to nearest
    even, subnormals kept) and the scale bookkeeping: what is left is the fp8 matrix core's internal accumulation (measured 7.6-8.0e-6 = 2^-17 on every shape, independent of K).  Also
    printed: the distance to the unquantised float64 result (the format's own error, ~3e-2)."""
    ops, F = S.ops, torch.nn.functional
    x = philox("f8.x", (n, cin, h, w)) * 3.0
    x8 = _e4m3(x * 8.0) / 8.0
    try:
        with ops.conv_precision("fp8"):
            if kind == "conv":
                wt = philox("f8.w", (cout, cin, ks, ks)) * 0.05
                Sw = _w_scale(wt)
                want = F.conv2d(x8, _e4m3(wt * Sw) / Sw, padding=ks // 2)
                exact = F.conv2d(x.double(), wt.double(), padding=ks // 2)
                y = torch.empty((n, cout, h, w), device=DEV)
                part = ops.conv2d(ops.full(g(x)), g(wt), None, ops.full(y), stats=True)
            else:
                wt = philox("f8.wt", (cin, cout, 2, 2)) * 0.05
                Sw = _w_scale(wt)
                want = F.conv_transpose2d(x8, _e4m3(wt * Sw) / Sw, stride=2)
                exact = F.conv_transpose2d(x.double(), wt.double(), stride=2)
                y = torch.empty((n, cout, 2 * h, 2 * w), device=DEV)
                part = ops.tconv2x2(ops.full(g(x)), g(wt), ops.full(y), stats=True)
            torch.cuda.synchronize()
        got = y.cpu().double()
        e, eq = rel_err(got, want), rel_err(got, exact)
        print(f"fp8 {kind} {cin}->{cout} @{h}x{w} ks={ks}: vs quantised float64 {e:.2e}; vs exact float64 {eq:.2e}")
        assert e < 3e-5, e                     # measured 7.6-8.0e-6 (one flipped e4m3 rounding would show as >= 1e-4)
        assert eq < 8e-2, eq                   # e4m3: 2^-4 per operand, measured ~3-4e-2
        # the fused plane statistics describe the stored output
        sc, sh = torch.empty((n, cout), device=DEV), torch.empty((n, cout), device=DEV)
        ops.norm_finalize(part, 0, 1e-5, sc, sh, 0)
        mu, var = got.mean((2, 3)), got.var((2, 3), unbiased=False)
        assert rel_err(sc.cpu().double(), 1.0 / torch.sqrt(var + 1e-5)) < 2e-5
        assert rel_err((sh / sc).cpu().double(), -mu) < 2e-4
        # back in the default mode the same call is fp32-equivalent again
        if kind == "conv":
            ops.conv2d(ops.full(g(x)), g(wt), None, ops.full(y))
            assert rel_err(y.cpu().double(), exact) < 3e-6
    finally:
        ops.set_conv_precision("bf16x3")


def test_fp8_lazy_affine_and_clamp(S):
    """[round 2] fp8 staging with the lazy InstanceNorm affine + LeakyReLU in front of the conversion, and activations beyond the
    e4m3 range (|8 a| > 448 saturates to +-448 instead of turning into NaN)."""
    ops, F = S.ops, torch.nn.functional
    n, cin, cout, h, w = 2, 36, 36, 32, 64
    x = philox("f8a.x", (n, cin, h, w)) * 2.0
    x[0, 3, 5, 7], x[1, 20, 9, 40] = 500.0, -300.0                     # outliers: 8 * lrelu(.) far outside +-448
    sc, sh = philox("f8a.sc", (n, cin), lo=0.5, hi=1.5), philox("f8a.sh", (n, cin))
    wt = philox("f8a.w", (cout, cin, 3, 3)) * 0.05
    try:
        with ops.conv_precision("fp8"):
            act = torch.empty((n, cin, h, w), device=DEV)
            ops.apply(ops.full(g(x), g(sc), g(sh), 0.2), ops.full(act))           # the device's own fp32 activation values
            y = torch.empty((n, cout, h, w), device=DEV)
            ops.conv2d(ops.full(g(x), g(sc), g(sh), 0.2), g(wt), None, ops.full(y))
            torch.cuda.synchronize()
        Sw = _w_scale(wt)
        want = F.conv2d(_e4m3(act.cpu() * 8.0) / 8.0, _e4m3(wt * Sw) / Sw, padding=1)
        assert torch.isfinite(y).all()
        e = rel_err(y.cpu().double(), want)
        print(f"fp8 lazy affine + clamp: vs quantised float64 {e:.2e}")
        assert e < 2e-4, e                    # measured 1.8e-5; a last-bit difference in the fp32 affine can flip single e4m3 roundings
    finally:
        ops.set_conv_precision("bf16x3")


def test_deferred_weight_gradient_reductions_are_bit_identical(S):
    """[round 2] san_wgrad_defer: inside wgrad_overlap the matrix-core weight gradients queue the fixed-order reduction of their partial
    tiles and one launch reduces up to 48 layers.  Two 'Rec' steps of an 18-channel model (> 48 queued layers per step, so
    the automatic flush is exercised) leave bit-identical parameters to the immediate form, and nothing stays queued."""
    ops = S.ops
    n, c, h, w = 2, 3, 48, 80

    def run(defer: bool):
        ops.WGRAD_DEFER[0] = defer
        try:
            cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                                weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=3, chans=18,
                                sens_chans=8, pools=2, sens_pools=2)
            net = S.model.CSModel(cfg)
            net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
            _load(S, net.net_T, 41)
            _load(S, net.net_R, 42)
            net.to(DEV).train()
            img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
            for _ in range(2):
                net.set_input(g(img_full), g(img_aux))
                net.update()
            torch.cuda.synchronize()
            assert ops.lib().query("san_wgrad_defer_pending") == 0
            assert ops.lib().query("san_wgrad_defer", 0) == 0            # update() leaves the mode off
            return [p.detach().cpu().clone() for m in (net.net_R, net.net_T) for p in m.parameters()]
        finally:
            ops.WGRAD_DEFER[0] = True

    a, b = run(True), run(False)
    assert all(torch.equal(x, y) for x, y in zip(a, b)), "deferred reductions change the result"


@pytest.mark.parametrize("cin,cout,h,w,bias", [(18, 18, 160, 192, False), (36, 18, 160, 160, False), (18, 36, 160, 160, False),
                                               (36, 36, 168, 160, True), (72, 36, 160, 160, False), (96, 32, 160, 160, True),
                                               (20, 16, 160, 160, False), (48, 48, 160, 160, False)])
def test_stream_convolution_vs_float64_and_tile_kernel(S, cin, cout, h, w, bias):
    """[round 4] conv3x3_stream_kernel (persistent workgroups; csrc/san_conv_stream.hip) on every template form, one to four 24-channel
    chunks, partial last chunks, a channel view with offset, bias, statistics: <= 3e-6 relative L2 of float64 (measured
    3.5-4.5e-7), statistics records that merge to the plane's mean / variance, and the same layer on the one-tile kernel
    (SAN_CONV_STREAM off via the tuning hook) within 1e-6 of it."""
    ops = S.ops
    n = 3
    assert S.lib.lib().query("san_conv_stream_eligible", n, h, w, cin, cout, cin + 3) == 1
    xb = g(philox("st.x", (n, cin + 3, h, w)))
    wt = g(philox("st.w", (cout, cin, 3, 3))) * 0.1
    bs = g(philox("st.b", (cout,))) if bias else None
    sc, sh = g(philox("st.sc", (n, cin + 3), lo=0.5, hi=1.5)), g(philox("st.sh", (n, cin + 3)))
    xa = ops.Act(xb, 2, cin, sc, sh, 0.2)
    yb = torch.full((n, cout + 2, h, w), 7.0, device=DEV)
    part = ops.conv2d(xa, wt, bs, ops.Act(yb, 1, cout), stats=True)
    torch.cuda.synchronize()
    want = _conv_ref64(xb[:, 2:2 + cin], sc[:, 2:2 + cin], sh[:, 2:2 + cin], 0.2, wt, bs)
    got = yb[:, 1:1 + cout]
    err = ((got.double() - want).norm() / want.norm()).item()
    assert err < 3e-6, err
    assert torch.all(yb[:, 0] == 7.0) and torch.all(yb[:, -1] == 7.0), "wrote outside its channel view"
    cnt, mean, m2 = part[..., 0].double(), part[..., 1].double(), part[..., 2].double()
    tot = cnt.sum(-1)
    mu = (cnt * mean).sum(-1) / tot
    var = (m2 + cnt * (mean - mu[..., None]) ** 2).sum(-1) / tot
    assert torch.all(tot == h * w)
    assert (mu - got.double().mean((2, 3))).abs().max().item() < 1e-6
    assert ((var - got.double().var((2, 3), unbiased=False)).abs() / got.double().var((2, 3), unbiased=False)).max().item() < 1e-5
    # the same layer on the one-tile-per-workgroup kernel
    S.lib.lib().call("san_conv_stream_set_tuning", 0)
    try:
        y2 = torch.full((n, cout + 2, h, w), 7.0, device=DEV)
        ops.conv2d(xa, wt, bs, ops.Act(y2, 1, cout), stats=True)
        torch.cuda.synchronize()
    finally:
        S.lib.lib().call("san_conv_stream_set_tuning", 1)
    assert ((got.double() - y2[:, 1:1 + cout].double()).norm() / want.norm()).item() < 1e-6


def test_stream_data_gradient_with_amax_scale(S):
    """[round 4] The stream kernel as the data gradient of a 3x3 convolution: dy of magnitude 1e-7 scaled by its recorded power of two
    (two fp16 parts), no input affine; <= 3e-6 of float64 (measured 3.8e-7)."""
    ops = S.ops
    n, cin, cout, h, w = 2, 36, 18, 160, 160
    wt = g(philox("sd.w", (cout, cin, 3, 3))) * 0.1
    gy = g(philox("sd.g", (n, cout, h, w))) * 1e-7
    ga = ops.full(gy)
    if ops.F16_BWD[0]:
        ga.amax = ops.amax_record(gy.abs().max())
    dx = torch.full((n, cin, h, w), float("nan"), device=DEV)
    ops.conv2d_dgrad(ga, wt, ops.full(dx))
    torch.cuda.synchronize()
    want = torch.nn.functional.conv_transpose2d(gy.double(), wt.double(), padding=1)
    assert ((dx.double() - want).norm() / want.norm()).item() < 3e-6


def test_stream_convolution_repeated_launches_are_deterministic(S):
    """[round 4] 200 launches of the persistent kernel on the same data: bit-identical outputs and statistics (and no hang: an early build
    with register spills stalled intermittently at three workgroups per CU)."""
    ops = S.ops
    n, cin, cout, h, w = 8, 18, 18, 320, 320
    xb, wt = g(philox("sr.x", (n, cin, h, w))), g(philox("sr.w", (cout, cin, 3, 3))) * 0.1
    sc, sh = g(philox("sr.sc", (n, cin), lo=0.5, hi=1.5)), g(philox("sr.sh", (n, cin)))
    xa = ops.Act(xb, 0, cin, sc, sh, 0.2)
    y0 = torch.empty((n, cout, h, w), device=DEV)
    p0 = ops.conv2d(xa, wt, None, ops.full(y0), stats=True).clone()
    y1 = torch.empty_like(y0)
    for _ in range(200):
        p1 = ops.conv2d(xa, wt, None, ops.full(y1), stats=True)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(p0, p1)


@pytest.mark.parametrize("cin,cout", [(18, 18), (36, 18), (18, 36), (72, 36)])
def test_stream_convolution_one_bf16_part(S, cin, cout):
    """[round 4] The persistent kernel's one-part form (san_set_conv_precision 'bf16': BASELINE configs[1] as written): activations and
    weights rounded to bf16, one product per MAC, fp32 accumulation -- against float64 on the SAME rounded operands (<= 2e-6:
    only the accumulation order differs) and against the one-tile kernel in the same mode; forward with lazy affine +
    statistics, and the data gradient."""
    ops = S.ops
    n, h, w = 2, 160, 160
    xb = g(philox("s1.x", (n, cin, h, w)))
    wt = g(philox("s1.w", (cout, cin, 3, 3))) * 0.1
    sc, sh = g(philox("s1.sc", (n, cin), lo=0.5, hi=1.5)), g(philox("s1.sh", (n, cin)))
    xa = ops.Act(xb, 0, cin, sc, sh, 0.2)
    with ops.conv_precision("bf16"):
        assert S.lib.lib().query("san_conv_stream_eligible", n, h, w, cin, cout, cin) == 1
        y = torch.full((n, cout, h, w), float("nan"), device=DEV)
        part = ops.conv2d(xa, wt, None, ops.full(y), stats=True)
        torch.cuda.synchronize()
        S.lib.lib().call("san_conv_stream_set_tuning", 0)
        try:
            y2 = torch.full((n, cout, h, w), float("nan"), device=DEV)
            ops.conv2d(xa, wt, None, ops.full(y2), stats=True)
            torch.cuda.synchronize()
        finally:
            S.lib.lib().call("san_conv_stream_set_tuning", 1)
        gy = g(philox("s1.g", (n, cout, h, w))) * 1e-6
        dx = torch.full((n, cin, h, w), float("nan"), device=DEV)
        ops.conv2d_dgrad(ops.full(gy), wt, ops.full(dx))
        torch.cuda.synchronize()
    a = xb.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
    a = torch.where(a >= 0, a, a * 0.2).float().bfloat16().double()
    want = torch.nn.functional.conv2d(a, wt.bfloat16().double(), padding=1)
    err = ((y.double() - want).norm() / want.norm()).item()
    assert err < 2e-6, err
    assert ((y.double() - y2.double()).norm() / want.norm()).item() < 1e-6
    cnt, mean = part[..., 0].double(), part[..., 1].double()
    assert torch.all(cnt.sum(-1) == h * w)
    assert ((cnt * mean).sum(-1) / cnt.sum(-1) - y.double().mean((2, 3))).abs().max().item() < 1e-6
    wantg = torch.nn.functional.conv_transpose2d(gy.bfloat16().double(), wt.bfloat16().double(), padding=1)
    assert ((dx.double() - wantg).norm() / wantg.norm()).item() < 2e-6


@pytest.mark.parametrize("n,cin,cout,h,w,ks,dgrad,aff,bias_on", [
    (8, 4, 18, 320, 320, 3, False, True, False),        # the cascade's first convolution (varnet.py:139-146)
    (8, 18, 2, 320, 320, 1, False, True, True),         # its output convolution
    (8, 2, 18, 320, 320, 1, True, False, False),        # ... and their data gradients
    (8, 18, 4, 320, 320, 3, True, False, False),
    (2, 3, 7, 50, 37, 3, False, True, True),            # odd sizes, W % 4 != 0
    (1, 2, 64, 96, 132, 3, False, False, True),         # several channel groups
    (3, 20, 3, 61, 70, 3, True, False, False),          # several input chunks, partial last chunk
    (2, 1, 8, 40, 23, 1, False, True, False),
    (15, 2, 8, 160, 92, 3, False, False, False)])       # the sensitivity net's first layer on 15 coil planes
def test_direct_small_channel_convolution_vs_float64_and_outer_product_kernel(S, n, cin, cout, h, w, ks, dgrad, aff, bias_on):
    """[round 5] conv_direct_kernel (csrc/san_conv_mfma.hip) behind san_conv2d_fwd: outputs and merged statistics against float64 (3e-6, the
    bar of the fp32 convolutions; measured 1-5e-7), and within 1e-6 of the outer-product kernel it replaces for these shapes."""
    gen = torch.Generator().manual_seed(11)
    x = g(torch.randn(n, cin, h, w, generator=gen))
    sc = g(torch.rand(n, cin, generator=gen) + 0.5) if aff else None
    sh = g(torch.randn(n, cin, generator=gen) * 0.3) if aff else None
    wt = g(torch.randn(*((cin, cout) if dgrad else (cout, cin)), ks, ks, generator=gen) * 0.1)
    bias = g(torch.randn(cout, generator=gen)) if bias_on else None
    y = torch.empty(n, cout, h, w, device=DEV)
    xa = S.ops.Act(x, 0, cin, sc, sh, 0.2 if aff else 1.0)
    xin = _act64(x, sc, sh, 0.2) if aff else x.double()
    if dgrad:
        want = F.conv2d(xin, wt.double().flip(2, 3).transpose(0, 1), padding=ks // 2)
    else:
        want = F.conv2d(xin, wt.double(), None if bias is None else bias.double(), padding=ks // 2)
    outs = {}
    try:
        for on in (True, False):
            S.ops.conv_direct(on)
            if dgrad:
                S.ops.conv2d_dgrad(xa, wt, S.ops.full(y))
                part = None
            else:
                part = S.ops.conv2d(xa, wt, bias, S.ops.full(y), stats=True, tag="t5").clone()
            torch.cuda.synchronize()
            outs[on] = (y.clone(), part)
    finally:
        S.ops.conv_direct(True)
    scale = want.abs().max()
    assert ((outs[True][0].double() - want).abs().max() / scale).item() < 3e-6
    assert ((outs[True][0] - outs[False][0]).abs().max() / scale).item() < 1e-6
    if not dgrad:
        mu, var, tot = _merge_stats(outs[True][1])
        assert float((tot - h * w).abs().max()) == 0.0
        assert ((mu - want.mean((2, 3))).abs().max() / scale).item() < 3e-6
        wv = want.var((2, 3), unbiased=False)
        assert ((var - wv).abs().max() / wv.max()).item() < 3e-6


@pytest.mark.parametrize("n,cin,cout,h,w", [(8, 288, 144, 20, 20), (8, 144, 72, 40, 40), (8, 72, 36, 80, 80), (8, 36, 18, 160, 160),
                                            (2, 40, 20, 23, 46), (1, 64, 16, 92, 160), (15, 16, 8, 160, 92)])
def test_transposed_convolution_as_one_stage_gemm(S, n, cin, cout, h, w):
    """[round 5] gemm1x1_f16_kernel (csrc/san_conv1x1.hip): ConvTranspose2d 2x2 s2 (varnet.py:159-192) forward with its statistics, and its
    data gradient on an amax-scaled gradient input, against float64 (3e-6; measured 2-6e-7) and the tiled kernel's KS = 1 form
    (1e-6).  The unused statistics slots must be empty, finite records."""
    gen = torch.Generator().manual_seed(12)
    x = g(torch.randn(n, cin, h, w, generator=gen))
    sc, sh = g(torch.rand(n, cin, generator=gen) + 0.5), g(torch.randn(n, cin, generator=gen) * 0.3)
    wt = g(torch.randn(cin, cout, 2, 2, generator=gen) * (1.0 / cin ** 0.5))
    xa = S.ops.Act(x, 0, cin, sc, sh, 0.2)
    y = torch.empty(n, cout, 2 * h, 2 * w, device=DEV)
    want = F.conv_transpose2d(_act64(x, sc, sh, 0.2), wt.double(), stride=2)
    dyp = g(torch.randn(n, 4 * cout, h, w, generator=gen) * 3e-5)
    rec = S.ops.AMAX.next(DEV)
    rec.zero_()
    rec.view(torch.float32)[0] = dyp.abs().max()
    da = S.ops.Act(dyp, 0, 4 * cout)
    da.amax = rec
    dx = torch.empty(n, cin, h, w, device=DEV)
    wv = wt.reshape(cin, 4 * cout, 1, 1)
    wantd = F.conv2d(dyp.double(), wv.double())
    res = {}
    try:
        for on in (True, False):
            S.ops.conv1x1_gemm(on)
            part = S.ops.tconv2x2(xa, wt, S.ops.full(y), stats=True, tag="t5").clone()
            S.ops.conv2d(da, wv, None, S.ops.full(dx), grad_input=True)
            torch.cuda.synchronize()
            res[on] = (y.clone(), part, dx.clone())
    finally:
        S.ops.conv1x1_gemm(True)
    scale = want.abs().max()
    assert ((res[True][0].double() - want).abs().max() / scale).item() < 3e-6
    assert ((res[True][0] - res[False][0]).abs().max() / scale).item() < 1e-6
    mu, var, tot = _merge_stats(res[True][1])
    assert float((tot - 4 * h * w).abs().max()) == 0.0
    assert ((mu - want.mean((2, 3))).abs().max() / scale).item() < 3e-6
    wvar = want.var((2, 3), unbiased=False)
    assert ((var - wvar).abs().max() / wvar.max()).item() < 3e-6
    assert ((res[True][2].double() - wantd).abs().max() / wantd.abs().max()).item() < 3e-6
    assert ((res[True][2] - res[False][2]).abs().max() / wantd.abs().max()).item() < 1e-6


@pytest.mark.parametrize("n,cin,cout,h,w", [(8, 64, 64, 160, 160), (2, 48, 40, 33, 50), (8, 128, 32, 80, 80)])
def test_1x1_convolution_as_one_stage_gemm_with_bias_views_and_statistics(S, n, cin, cout, h, w):
    """[round 5] The plain 1x1 form (unet.py's 1x1 layers): channel views on both sides, bias, statistics; HW % 4 != 0 takes the scalar stores."""
    gen = torch.Generator().manual_seed(13)
    xb = g(torch.randn(n, cin + 5, h, w, generator=gen))
    sc, sh = g(torch.rand(n, cin + 5, generator=gen) + 0.5), g(torch.randn(n, cin + 5, generator=gen) * 0.3)
    wt = g(torch.randn(cout, cin, 1, 1, generator=gen) * (1.0 / cin ** 0.5))
    bias = g(torch.randn(cout, generator=gen))
    want = F.conv2d(_act64(xb[:, 2:2 + cin], sc[:, 2:2 + cin], sh[:, 2:2 + cin], 0.2), wt.double(), bias.double())
    xa = S.ops.Act(xb, 2, cin, sc, sh, 0.2)
    try:
        for on in (True, False):
            S.ops.conv1x1_gemm(on)
            yb = torch.zeros(n, cout + 3, h, w, device=DEV)
            part = S.ops.conv2d(xa, wt, bias, S.ops.Act(yb, 1, cout), stats=True, tag="u5")
            torch.cuda.synchronize()
            assert float(yb[:, 0].abs().max()) == 0.0 and float(yb[:, 1 + cout:].abs().max()) == 0.0      # nothing outside the view
            scale = want.abs().max()
            assert ((yb[:, 1:1 + cout].double() - want).abs().max() / scale).item() < 3e-6
            mu, var, tot = _merge_stats(part)
            assert float((tot - h * w).abs().max()) == 0.0
            assert ((mu - want.mean((2, 3))).abs().max() / scale).item() < 3e-6
            wvar = want.var((2, 3), unbiased=False)
            assert ((var - wvar).abs().max() / wvar.max()).item() < 3e-6
    finally:
        S.ops.conv1x1_gemm(True)


def test_one_stage_gemm_plain_bf16_form_matches_the_tiled_kernel(S):
    """[round 5] Narrow-precision mode (one bf16 part): the GEMM form of a transposed convolution and of its data gradient against the tiled
    kernel's (same roundings, same accumulation order per output: <= 1e-6 of each other) and against float64 at bf16 level."""
    n, cin, cout, h, w = 8, 72, 36, 80, 80
    gen = torch.Generator().manual_seed(14)
    x = g(torch.randn(n, cin, h, w, generator=gen))
    sc, sh = g(torch.rand(n, cin, generator=gen) + 0.5), g(torch.randn(n, cin, generator=gen) * 0.3)
    wt = g(torch.randn(cin, cout, 2, 2, generator=gen) * (1.0 / cin ** 0.5))
    xa = S.ops.Act(x, 0, cin, sc, sh, 0.2)
    y = torch.empty(n, cout, 2 * h, 2 * w, device=DEV)
    want = F.conv_transpose2d(_act64(x, sc, sh, 0.2), wt.double(), stride=2)
    dyp = g(torch.randn(n, 4 * cout, h, w, generator=gen) * 3e-5)
    dx = torch.empty(n, cin, h, w, device=DEV)
    wv = wt.reshape(cin, 4 * cout, 1, 1)
    wantd = F.conv2d(dyp.double(), wv.double())
    res = {}
    try:
        with S.ops.conv_precision("bf16"):
            for on in (True, False):
                S.ops.conv1x1_gemm(on)
                part = S.ops.tconv2x2(xa, wt, S.ops.full(y), stats=True, tag="b5").clone()
                S.ops.conv2d(S.ops.Act(dyp, 0, 4 * cout), wv, None, S.ops.full(dx), grad_input=True)
                torch.cuda.synchronize()
                res[on] = (y.clone(), part, dx.clone())
    finally:
        S.ops.conv1x1_gemm(True)
    scale = want.abs().max()
    assert ((res[True][0] - res[False][0]).abs().max() / scale).item() < 1e-6
    assert ((res[True][2] - res[False][2]).abs().max() / wantd.abs().max()).item() < 1e-6
    assert ((res[True][0].double() - want).abs().max() / scale).item() < 2e-2
    assert ((res[True][2].double() - wantd).abs().max() / wantd.abs().max()).item() < 2e-2
    mu, var, tot = _merge_stats(res[True][1])
    assert float((tot - 4 * h * w).abs().max()) == 0.0


def test_round5_kernels_repeat_bit_identically(S):
    """[round 5] 100 launches each of the direct kernel (4 -> 18, 3x3, statistics) and of the GEMM form (transposed convolution 72 -> 36 with
    statistics, its data gradient with an amax scale) give identical bits every time (no atomics, fixed reduction orders)."""
    gen = torch.Generator().manual_seed(15)
    n = 8
    x4 = g(torch.randn(n, 4, 160, 160, generator=gen))
    w4 = g(torch.randn(18, 4, 3, 3, generator=gen) * 0.1)
    y4 = torch.empty(n, 18, 160, 160, device=DEV)
    x = g(torch.randn(n, 72, 80, 80, generator=gen))
    sc, sh = g(torch.rand(n, 72, generator=gen) + 0.5), g(torch.randn(n, 72, generator=gen) * 0.3)
    wt = g(torch.randn(72, 36, 2, 2, generator=gen) * 0.1)
    y = torch.empty(n, 36, 160, 160, device=DEV)
    dyp = g(torch.randn(n, 144, 80, 80, generator=gen) * 1e-4)
    rec = S.ops.AMAX.next(DEV)
    rec.zero_()
    rec.view(torch.float32)[0] = dyp.abs().max()
    da = S.ops.Act(dyp, 0, 144)
    da.amax = rec
    dx = torch.empty(n, 72, 80, 80, device=DEV)
    wv = wt.reshape(72, 144, 1, 1)
    first = None
    for it in range(100):
        p4 = S.ops.conv2d(S.ops.full(x4), w4, None, S.ops.full(y4), stats=True, tag="rep4")
        pt = S.ops.tconv2x2(S.ops.Act(x, 0, 72, sc, sh, 0.2), wt, S.ops.full(y), stats=True, tag="rept")
        S.ops.conv2d(da, wv, None, S.ops.full(dx), grad_input=True)
        cur = [t.clone() for t in (y4, p4, y, pt, dx)]
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, cur)), it


# ------------------------------------------------------------------ round 6: operand RANGE of the fp16 two-part forward format
@pytest.mark.parametrize("n,cin,cout,h,w,ks", [(2, 72, 72, 40, 40, 3), (1, 18, 18, 320, 320, 3), (2, 64, 64, 16, 32, 1), (1, 288, 144, 20, 20, 3)])
@pytest.mark.parametrize("ws", [1e-4, 1.0, 50.0])
@pytest.mark.parametrize("xs", [1e-5, 1.0, 3e4])
def test_f16x2_forward_operand_ranges(S, n, cin, cout, h, w, ks, xs, ws):
    """[round 6] The FORWARD fp16 two-part form over the operand ranges, on the tiled, persistent, one-stage GEMM and
    split-K kernels; the input is read through a lazy affine that brings an O(1) tensor to magnitude `xs`, the weights are `ws`-sized.

    WEIGHTS: fp16's range as it is by default (trained weights are O(0.01 - 1): every element then keeps >= 19 bits and the layer
    3e-7); with the per-tensor power-of-two scale switched on (ops.f16_weight_scale(True) / SAN_F16_WSCALE=1: the packed image
    holds w S_w, max |w| S_w in [2^13, 2^14), the accumulators get 1 / S_w) weights of ANY magnitude are exact to 22 bits --
    asserted for 1e-4-sized and 50-sized tensors.  ACTIVATIONS have fp16's range as it is (normalised tensors are O(1) by
    construction, images are <= 1): at |x| = 1e-5 the second part sits in the denormals and the format keeps only an ABSOLUTE
    accuracy of 2^-25 per element -- asserted as such -- and the documented way out is the three-part bf16 form (fp32's exponent
    range: ops.F16_FWD[0] = False / SAN_NO_F16X2=1), which holds 3e-6 there too."""
    ops = S.ops
    x = philox("rg.x", (n, cin, h, w))                  # (|activated input| <= 1.25 xs: 3.75e4 at the top, inside fp16's 65504)
    wt = philox("rg.w", (cout, cin, ks, ks)) * ws
    sc = torch.full((n, cin), xs)
    sh = torch.full((n, cin), 0.25 * xs)
    act = lambda t: torch.nn.functional.leaky_relu(t * xs + 0.25 * xs, 0.2)     # noqa: E731
    y64 = torch.nn.functional.conv2d(act(x.double()), wt.double(), padding=ks // 2)

    def run():
        y = torch.empty((n, cout, h, w), device=DEV)
        ops.conv2d(ops.Act(g(x), 0, cin, g(sc), g(sh), 0.2), g(wt), None, ops.full(y))
        torch.cuda.synchronize()
        return y.cpu().double()

    assert ops.F16_FWD[0] and ops.lib().query("san_get_conv_precision") == 3
    prev = ops.f16_weight_scale(ws != 1.0)              # (O(1) weights: the default, unscaled format)
    try:
        e16 = rel_err(run(), y64)
        y16 = run()
    finally:
        ops.f16_weight_scale(prev)
    if xs >= 1.0:
        assert e16 < 3e-6, (xs, ws, e16)
    else:
        # the activations' absolute floor: |error| <= 2^-25 per element of the lazily activated input, times the weights it meets
        bound = 2.0 ** -25 * wt.double().abs().sum(dim=(1, 2, 3)).max().item()
        assert (y16 - y64).abs().max().item() <= 1.5 * bound
        ops.F16_FWD[0] = False
        try:
            e3 = rel_err(run(), y64)
        finally:
            ops.F16_FWD[0] = True
        assert e3 < 3e-6, (xs, ws, e3)


# ------------------------------------------------------------------ round 6: launch plans of the 40^2 level
@pytest.mark.parametrize("n,cin,cout,h,w", [
    (8, 72, 144, 40, 40),       # 240 workgroups of 40 x 4 tiles (three blocks per wave), weights through LDS
    (8, 144, 144, 40, 40),
    (4, 144, 72, 40, 40),       # the decoder's data-gradient shape; split over K
    (8, 72, 36, 80, 80),        # 32 x 8 tiles, 240 workgroups: weights through LDS
    (3, 40, 24, 23, 44),        # ragged: 44-wide rows, 23 rows, partial channel blocks
    (2, 144, 288, 20, 20),      # the 20^2 level keeps its 256-pixel tiles
])
def test_short_tiles_and_lds_staged_weights_match_the_round5_forms_bitwise(S, n, cin, cout, h, w):
    """[round 6] The launch plan of a 3x3 convolution -- short full-width tiles (three 16-pixel blocks per wave) where the default
    leaves compute units idle, weights staged through LDS where a launch has at most one workgroup per unit (san_conv_bf16.hip:
    tile_plan, g_b16_wd_cold) -- changes WHICH workgroup computes a pixel, not the order of its sum: outputs are bit-identical to
    the round-5 forms (256-pixel tiles, weights direct), forward and data gradient; both within 3e-6 of float64; the statistics
    (other tiles, other merge order) finalise to the same InstanceNorm affine within 2e-6.  (nn.Conv2d + InstanceNorm2d,
    varnet.py:139-146.)"""
    ops = S.ops
    x = g(philox("r6p.x", (n, cin, h, w)) * 2)
    wt = g(philox("r6p.w", (cout, cin, 3, 3)) * 0.1)
    sc = g(philox("r6p.sc", (n, cin)).abs() + 0.5)
    sh = g(philox("r6p.sh", (n, cin)) * 0.3)
    gout = g(philox("r6p.g", (n, cout, h, w)) * 1e-3)
    xa = F.leaky_relu(x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None], 0.2)
    y64 = F.conv2d(xa, wt.double(), padding=1)
    dx64 = F.conv_transpose2d(gout.double(), wt.double(), padding=1)
    res = []
    try:
        for nbw, wd_cold in ((0, 1), (4, 0)):
            ops.lib().call("san_conv_bf16x3_tile_set_tuning", nbw, wd_cold)
            y = ops.Act(torch.zeros((n, cout, h, w), device=DEV), 0, cout, torch.zeros((n, cout), device=DEV), torch.zeros((n, cout), device=DEV), 0.2)
            part = ops.conv2d(ops.Act(x, 0, cin, sc, sh, 0.2), wt, None, y, stats=True, tag=f".r6p{nbw}", instance_norm_eps=1e-5)
            if part is not None:
                ops.norm_finalize(part, ops.NORM_INSTANCE, 1e-5, y.scale, y.shift, y.coff)
            ops.AMAX.reset(DEV)
            dy = ops.Act(torch.empty((n, cout, h, w), device=DEV), 0, cout)
            ops.act_bwd(ops.full(gout), ops.full(torch.ones_like(gout)), dy, instance_norm=False)
            dx = torch.empty((n, cin, h, w), device=DEV)
            ops.conv2d_dgrad(dy, wt, ops.full(dx))
            torch.cuda.synchronize()
            res.append((y.buf.clone(), y.scale.clone(), y.shift.clone(), dx.clone()))
    finally:
        ops.lib().call("san_conv_bf16x3_tile_set_tuning", 0, 1)
    (ya, sca, sha, dxa), (yb, scb, shb, dxb) = res
    assert torch.equal(ya, yb) and torch.equal(dxa, dxb)
    assert rel_err(ya.cpu().double(), y64.cpu()) < 3e-6 and rel_err(dxa.cpu().double(), dx64.cpu()) < 3e-6
    var, mean = torch.var_mean(y64, dim=(2, 3), unbiased=False)
    want_sc = torch.rsqrt(var + 1e-5)
    for got_sc, got_sh in ((sca, sha), (scb, shb)):
        assert ((got_sc.double() - want_sc).abs() / want_sc).max().item() < 2e-6
        assert ((got_sh.double() + mean * want_sc).abs().max() / (mean * want_sc).abs().max()).item() < 1e-5


@pytest.mark.parametrize("n,cin,cout,h,w,scale", [(8, 3, 18, 320, 320, 1e-3), (2, 2, 18, 160, 160, 3e-7), (1, 3, 36, 160, 192, 2e4), (2, 3, 18, 48, 64, 1.0)])
def test_data_gradient_to_two_or_three_channels_on_the_persistent_kernel(S, n, cin, cout, h, w, scale):
    """[round 6] dL/dx of Conv2d(cin = 2 / 3, cout, 3x3) -- a cascade's input convolution (varnet.py:139-146, in_chans 2 / 3) -- runs
    on the persistent matrix-core kernel with the 2 / 3 output channels as a partial block alone (conv3x3_stream_kernel<0, REM>) where
    san_conv_stream_eligible takes the shape, on the direct fp32 kernel elsewhere (the 48 x 64 case): <= 3e-6 of float64 either way,
    gradients of magnitude 3e-7 .. 2e4 (scaled by their recorded maximum), every other channel of the destination untouched."""
    ops = S.ops
    wt = g(philox("sc.w", (cout, cin, 3, 3)) * 0.1)
    gout = g(philox("sc.g", (n, cout, h, w)) * scale)
    ops.AMAX.reset(DEV)
    dy = ops.Act(torch.empty((n, cout, h, w), device=DEV), 0, cout)
    ops.act_bwd(ops.full(gout), ops.full(torch.ones_like(gout)), dy, instance_norm=False)
    taken = bool(ops.STREAM_SMALL_COUT[0] and ops.lib().query("san_conv_stream_eligible", n, h, w, cout, cin, cout))
    assert taken == (h * w >= 160 * 160)
    dst = torch.full((n, cin + 2, h, w), -3.0, device=DEV)
    ops.conv2d_dgrad(dy, wt, ops.Act(dst, 1, cin))
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(gout.double(), wt.double(), padding=1)
    assert rel_err(dst[:, 1:1 + cin].cpu().double(), ref.cpu()) < 3e-6
    assert float(dst[:, 0].min()) == -3.0 == float(dst[:, 0].max()) and float(dst[:, -1].min()) == -3.0 == float(dst[:, -1].max())


@pytest.mark.parametrize("n,cin,cout,h,w", [(8, 32, 2, 320, 320), (2, 18, 3, 160, 160), (1, 32, 2, 48, 64)])
def test_convolution_to_two_or_three_channels_on_the_persistent_kernel(S, n, cin, cout, h, w):
    """[round 6] Conv2d(cin >= 16, 2 / 3, 3x3) + bias without statistics -- the alignment net's last layer (unet.py:108-111: 32 -> 2,
    the displacement field) -- on the persistent matrix-core kernel (partial channel block alone; 69.9 -> 50.8 us at 32 -> 2 @320^2)
    where san_conv_stream_eligible takes the shape, else on the direct fp32 kernel: <= 3e-6 of float64, lazily activated input,
    a channel view as destination."""
    ops = S.ops
    x = g(philox("sf.x", (n, cin, h, w)) * 2)
    wt, b = g(philox("sf.w", (cout, cin, 3, 3)) * 0.1), g(philox("sf.b", (cout,)))
    sc, sh = g(philox("sf.sc", (n, cin)).abs() + 0.5), g(philox("sf.sh", (n, cin)) * 0.3)
    dst = torch.full((n, cout + 2, h, w), 5.0, device=DEV)
    ops.conv2d(ops.Act(x, 0, cin, sc, sh, 0.2), wt, b, ops.Act(dst, 0, cout))
    torch.cuda.synchronize()
    xa = F.leaky_relu(x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None], 0.2)
    ref = F.conv2d(xa, wt.double(), b.double(), padding=1)
    assert rel_err(dst[:, :cout].cpu().double(), ref.cpu()) < 3e-6
    assert float(dst[:, cout:].min()) == 5.0 == float(dst[:, cout:].max())
