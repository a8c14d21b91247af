"""GPU parity tests (through the C ABI), component: normalisation, activation and the element-wise materialisers (rows a8, a9, a11).
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


def test_group_norm_and_elementwise(S, ops_golden):
    """[round 1]"""
    x2 = philox("nu.x", (3, 2, 32, 48)) * 3 + 0.7
    xa = S.ops.Act(g(x2), 0, 2, torch.empty((3, 2), device=DEV), torch.empty((3, 2), device=DEV), 1.0)
    std = torch.empty((3, 2), device=DEV)
    mean = torch.empty((3, 2), device=DEV)
    S.ops.norm_finalize(S.ops.plane_stats(xa), S.ops.NORM_GROUP, 1e-6, xa.scale, xa.shift, 0, aux_a=std, aux_b=mean)
    assert rel_err(mean.cpu().view(3, 2, 1, 1), as_t(ops_golden["norm_mean"])) < 2e-6
    assert rel_err(std.cpu().view(3, 2, 1, 1), as_t(ops_golden["norm_std"])) < 2e-6
    y = torch.empty_like(xa.buf)
    S.ops.apply(xa, S.ops.full(y))
    assert rel_err(y.cpu(), as_t(ops_golden["norm_x"])) < 5e-6
    # avgpool / upsample / add with lazy affines
    a = philox("ew.a", (2, 5, 16, 24))
    sc, sh = philox("ew.sc", (2, 5), lo=0.5, hi=1.5), philox("ew.sh", (2, 5))
    act = lambda t: torch.nn.functional.leaky_relu(t * sc[:, :, None, None] + sh[:, :, None, None], 0.01)
    A = S.ops.Act(g(a), 0, 5, g(sc), g(sh), 0.01)
    y = torch.empty((2, 5, 8, 12), device=DEV)
    S.ops.avgpool2(A, S.ops.full(y))
    assert rel_err(y.cpu(), torch.nn.functional.avg_pool2d(act(a), 2)) < 1e-6
    y = torch.empty((2, 7, 32, 48), device=DEV)
    S.ops.upsample2(A, S.ops.Act(y, 2, 5))
    assert rel_err(y[:, 2:7].cpu(), torch.nn.functional.interpolate(act(a), scale_factor=2, mode="nearest")) < 1e-6
    y = torch.empty((2, 5, 16, 24), device=DEV)
    S.ops.add(A, S.ops.full(g(a)), S.ops.full(y))
    assert rel_err(y.cpu(), act(a) + a) < 1e-6


def test_instance_norm_act_backward(S):
    """[round 1]"""
    n, c, h, w = 2, 5, 24, 40
    y = philox("ib.y", (n, c, h, w)) * 2 + 0.3
    gout = philox("ib.g", (n, c, h, w))
    y64 = y.double().requires_grad_(True)
    mean = y64.mean(dim=(2, 3), keepdim=True)
    var = y64.var(dim=(2, 3), unbiased=False, keepdim=True)
    a = torch.nn.functional.leaky_relu((y64 - mean) / torch.sqrt(var + 1e-5), 0.2)
    a.backward(gout.double())
    # forward lazy affine through the library (plane stats -> finalize)
    ya = S.ops.Act(g(y), 0, c, torch.empty((n, c), device=DEV), torch.empty((n, c), device=DEV), 0.2)
    S.ops.norm_finalize(S.ops.plane_stats(ya), S.ops.NORM_INSTANCE, 1e-5, ya.scale, ya.shift, 0)
    dy = torch.empty((n, c, h, w), device=DEV)
    S.ops.act_bwd(S.ops.full(g(gout)), ya, S.ops.full(dy), instance_norm=True)
    assert rel_err(dy.cpu(), y64.grad.float()) < 2e-5
    # plain affine mode
    sc, sh = philox("ib.sc", (n, c), lo=0.5, hi=1.5), philox("ib.sh", (n, c))
    y64 = y.double().requires_grad_(True)
    torch.nn.functional.leaky_relu(y64 * sc[:, :, None, None].double() + sh[:, :, None, None].double(), 0.01).backward(gout.double())
    S.ops.act_bwd(S.ops.full(g(gout)), S.ops.Act(g(y), 0, c, g(sc), g(sh), 0.01), S.ops.full(dy), instance_norm=False)
    assert rel_err(dy.cpu(), y64.grad.float()) < 2e-6


# ------------------------------------------------------------------ window copy / padding
def test_window_copy_modes(S):
    """[round 2] san_window_copy_fwd against F.pad: zero pad, crop, reflect (bottom / right) and the reflect adjoint.  Exact."""
    F = torch.nn.functional
    x = philox("wc.x", (2, 5, 9, 13))
    sc, sh = philox("wc.sc", (2, 5), lo=0.5, hi=1.5), philox("wc.sh", (2, 5))
    act = F.leaky_relu(x * sc[:, :, None, None] + sh[:, :, None, None], 0.2)
    y = torch.empty((2, 5, 16, 16), device=DEV)
    S.ops.window_copy(S.ops.Act(g(x), 0, 5, g(sc), g(sh), 0.2), S.ops.full(y), 3, 1)
    want = F.pad(act, [1, 2, 3, 4])
    assert torch.allclose(y.cpu(), want, rtol=0, atol=1e-6)
    assert torch.equal(y.cpu() == 0, want == 0)                                     # the frame is exactly zero
    back = torch.empty((2, 5, 9, 13), device=DEV)
    S.ops.window_copy(S.ops.full(y), S.ops.full(back), -3, -1)
    assert torch.equal(back.cpu(), y.cpu()[:, :, 3:12, 1:14])
    for dh, dw in ((1, 1), (0, 1), (1, 0)):
        r = torch.empty((2, 5, 9 + dh, 13 + dw), device=DEV)
        S.ops.window_copy(S.ops.full(g(x)), S.ops.full(r), mode=1)
        assert torch.equal(r.cpu(), F.pad(x, [0, dw, 0, dh], "reflect"))
        gr = philox("wc.g", (2, 5, 9 + dh, 13 + dw))
        x64 = x.double().requires_grad_(True)
        F.pad(x64, [0, dw, 0, dh], "reflect").backward(gr.double())
        gx = torch.empty((2, 5, 9, 13), device=DEV)
        S.ops.window_copy(S.ops.full(g(gr)), S.ops.full(gx), mode=2)
        assert torch.allclose(gx.cpu().double(), x64.grad, rtol=0, atol=1e-6)
    # channel views on both sides
    big = torch.zeros((2, 8, 10, 14), device=DEV)
    S.ops.window_copy(S.ops.Act(g(x), 1, 3), S.ops.Act(big, 4, 3), mode=1)
    assert torch.equal(big.cpu()[:, 4:7], F.pad(x[:, 1:4], [0, 1, 0, 1], "reflect")) and big[:, :4].abs().sum().item() == 0


def test_normunet_backward_with_constant_plane(S):
    """[round 2] An all-zero slice in the batch has std == 0 on both planes: the reference stays finite (forward divides by
    std + 1e-6, torch's std backward masks std == 0); so must the hand-written backward.  Against oracle autograd."""
    n, h, w = 2, 32, 48
    net = S.varnet.NormUnet(4, 2, use_ref=True)
    params = _load(S, net, 78)
    net.to(DEV)
    x = cplx("nb.x", (n, 1, h, w)) * 2 + 0.5
    x[1] = 0
    ref = philox("nb.ref", (n, 1, h, w), lo=0.0, hi=1.0)
    gout = cplx("nb.g", (n, 1, h, w))
    p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
    x64 = x.to(torch.complex128).requires_grad_(True)
    y64 = S.O.normunet_forward(p64, "", x64, ref.double(), 2, True)
    (y64.real * gout.real.double() + y64.imag * gout.imag.double()).sum().backward()
    xin = net.input_buffer(n, h, w, DEV, "nbz")
    S.ops.apply(S.ops.full(g(torch.cat([x.real, x.imag], 1))), xin.view(0, 2))
    net.set_ref(xin, g(ref))
    out = torch.empty((n, 2, h, w), device=DEV)
    net.run(xin, out, "nbz")
    g_m, _ = net.run_bwd(g(torch.cat([gout.real, gout.imag], 1)), "nbz", want_ref_grad=False)
    assert torch.isfinite(g_m).all()
    want = torch.cat([x64.grad.real, x64.grad.imag], 1).float()
    assert rel_err(g_m[0].cpu(), want[0]) < 2e-4
    # the constant slice: d/dm of (m - mu)/(0 + 1e-6) is huge but finite; compare relative to its own scale
    assert rel_err(g_m[1].cpu(), want[1]) < 2e-3
    for name, prm in net.named_parameters():
        assert torch.isfinite(prm.grad).all(), name
        wantp = p64[name].grad.float()
        assert (prm.grad.cpu() - wantp).abs().max().item() <= 1e-3 * max(wantp.abs().max().item(), 1e-12), name


@pytest.mark.parametrize("n,c,h,w,coff", [(2, 5, 320, 320, 0), (2, 6, 160, 160, 1), (1, 3, 40, 24, 0), (3, 4, 6, 8, 2)])
def test_act_bwd_second_gradient_source(S, n, c, h, w, coff):
    """[round 2] san_act_bwd_up_amax: InstanceNorm + LeakyReLU backward whose incoming gradient is g + 0.25 * nearest_up(g2) (the
    U-Net encoder's skip gradient + avg_pool2d adjoint, varnet.py:118-134 under autograd), against float64 autograd of
    the composed expression; both kernel forms (one-pass planes up to 160 x 160, two-kernel above)."""
    ops = S.ops
    ct = c + coff + 1
    gsk, g2 = philox("abu.g", (n, ct, h, w)), philox("abu.g2", (n, ct, h // 2, w // 2)) * 3.0
    y = philox("abu.y", (n, ct, h, w)) * 2.0 + 0.3
    yd = y[:, coff:coff + c].double().requires_grad_(True)
    mu, var = yd.mean((2, 3), keepdim=True), yd.var((2, 3), unbiased=False, keepdim=True)
    a = torch.nn.functional.leaky_relu((yd - mu) / torch.sqrt(var + 1e-5), 0.2)
    gt = gsk[:, coff:coff + c].double() + 0.25 * torch.nn.functional.interpolate(g2[:, coff:coff + c].double(), scale_factor=2, mode="nearest")
    (want,) = torch.autograd.grad(a, yd, gt)
    sc = (1.0 / torch.sqrt(var + 1e-5)).reshape(n, c).float()
    sh = (-mu.reshape(n, c).double() * sc.double()).float()
    scf, shf = torch.ones(n, ct), torch.zeros(n, ct)
    scf[:, coff:coff + c], shf[:, coff:coff + c] = sc, sh
    dy = torch.zeros(n, ct, h, w, device=DEV)
    ops.AMAX.reset(torch.device(DEV))
    dya = ops.Act(dy, coff, c)
    ops.act_bwd(ops.Act(g(gsk), coff, c), ops.Act(g(y), coff, c, g(scf), g(shf), 0.2), dya, instance_norm=True,
                g2=ops.Act(g(g2), coff, c))
    got = dy[:, coff:coff + c].cpu().double()
    assert rel_err(got, want) < 5e-6, rel_err(got, want)
    assert dy[:, :coff].abs().sum().item() == 0.0 and dy[:, coff + c:].abs().sum().item() == 0.0      # the view's neighbours
    if dya.amax is not None:                               # the recorded maximum is the largest |dy| written
        assert abs(ops.amax_value(dya.amax) - dy.abs().max().item()) <= 1e-6 * dy.abs().max().item()


# ------------------------------------------------------------------ late round-2 entry points
def test_group_norm_backward_aux_and_partials_add(S):
    """[round 2] SAN_NORM_GROUP_BWD: the NormUnet statistics launch also writes the two per-plane values its backward needs (guarded
    1 / std and -mean / std; 0 for a constant plane, as torch's std backward masks it), and san_partials_add accumulates a
    scalar gradient from per-workgroup partials in double."""
    ops = S.ops
    n, h, w = 3, 40, 56
    x = philox("gnb.x", (n, 2, h, w)) * 2.0 + 0.5
    x[1, 0] = 0.75                                        # a constant plane: std == 0
    part = ops.plane_stats(ops.full(g(x)), tag="gnb")
    sc, sh = torch.empty((n, 2), device=DEV), torch.empty((n, 2), device=DEV)
    std2, mean2 = torch.empty((2, n, 2), device=DEV), torch.empty((2, n, 2), device=DEV)
    ops.norm_finalize(part, ops.NORM_GROUP_BWD, 1e-6, sc, sh, 0, aux_a=std2, aux_b=mean2)
    xd = x.double()
    std, mean = xd.std((2, 3)), xd.mean((2, 3))
    assert rel_err(std2[0].cpu().double(), std) < 1e-6 and rel_err(mean2[0].cpu().double(), mean) < 1e-6
    isd = torch.where(std > 1e-12, 1.0 / std.clamp_min(1e-30), torch.zeros_like(std))
    assert std2[0, 1, 0].item() == 0.0 and std2[1, 1, 0].item() == 0.0 and mean2[1, 1, 0].item() == 0.0
    assert rel_err(std2[1].cpu().double(), isd) < 1e-6 and rel_err(mean2[1].cpu().double(), -mean * isd) < 1e-6
    assert rel_err(sc.cpu().double(), 1.0 / (std + 1e-6)) < 1e-6
    # partials -> scalar gradient
    p = philox("gnb.p", (641,)) * 3.0
    dst = torch.full((1,), 0.25, device=DEV)
    ops.lib().call("san_partials_add", ops._p(g(p)), 641, -1.0, ops._p(dst), ops._stream())
    torch.cuda.synchronize()
    assert abs(dst.item() - (0.25 - p.double().sum().item())) < 1e-5


def test_splitk_instance_norm_finalised_in_the_reduction_is_bit_identical(S):
    """[round 2] A split-K convolution followed by InstanceNorm writes the lazy affine in its reduction pass
    (san_conv2d_bf16x3_fwd_ws_in): same bits as the separate san_norm_finalize launch on the partials it replaces; a layer
    that is not split still returns its partials."""
    ops = S.ops
    n, cin, cout, h, w = 2, 288, 288, 20, 20
    x, wt = philox("skin.x", (n, cin, h, w)), philox("skin.w", (cout, cin, 3, 3)) * 0.03
    assert ops.lib().query("san_conv_bf16x3_ws_bytes", n, h, w, cin, cout, 3) > 0           # this shape is split over K

    def run(eps):
        y = ops.Act(torch.empty((n, cout, h, w), device=DEV), 0, cout, torch.zeros((n, cout), device=DEV),
                    torch.zeros((n, cout), device=DEV), 0.2)
        part = ops.conv2d(ops.full(g(x)), g(wt), None, y, stats=True, instance_norm_eps=eps)
        if eps is None:
            assert part is not None
            ops.norm_finalize(part, ops.NORM_INSTANCE, 1e-5, y.scale, y.shift, 0)
        else:
            assert part is None                                                              # finalised in the reduction
        torch.cuda.synchronize()
        return y.buf.cpu(), y.scale.cpu(), y.shift.cpu()

    ya, sa, ha = run(None)
    yb, sb, hb = run(1e-5)
    assert torch.equal(ya, yb) and torch.equal(sa, sb) and torch.equal(ha, hb)
    yd = ya.double()
    assert rel_err(sa.double(), 1.0 / torch.sqrt(yd.var((2, 3), unbiased=False) + 1e-5)) < 1e-5
    # not split: 18 -> 18 at 64 x 64 keeps the partials + finalising launch
    x2, w2 = philox("skin.x2", (2, 18, 64, 64)), philox("skin.w2", (18, 18, 3, 3)) * 0.1
    y2 = ops.Act(torch.empty((2, 18, 64, 64), device=DEV), 0, 18, torch.zeros((2, 18), device=DEV), torch.zeros((2, 18), device=DEV), 0.2)
    assert ops.conv2d(ops.full(g(x2)), g(w2), None, y2, stats=True, instance_norm_eps=1e-5) is not None


@pytest.mark.parametrize("n,c,h,w", [(2, 5, 16, 24), (1, 3, 320, 320), (2, 4, 40, 40)])
def test_act_bwd_destination_modes_are_bit_identical(S, n, c, h, w):
    """[round 3] san_act_bwd_ex_amax: the pixel-unshuffled store equals san_act_bwd_amax + san_unshuffle2_fwd bit for bit (one-pass plane
    kernel and the two-kernel form for 320 x 320), and the accumulate form equals a separate add."""
    ops = S.ops
    gv, yv = g(philox("abx.g", (n, c, h, w))), g(philox("abx.y", (n, c, h, w)))
    sc, sh = g(philox("abx.sc", (n, c), lo=0.5, hi=1.5)), g(philox("abx.sh", (n, c)))
    ya = ops.Act(yv, 0, c, sc, sh, 0.2)
    ref_dy = torch.empty((n, c, h, w), device=DEV)
    ops.act_bwd(ops.full(gv), ya, ops.full(ref_dy), instance_norm=True)
    want = torch.empty((n, 4 * c + 3, h // 2, w // 2), device=DEV).fill_(7.0)
    ops.unshuffle2(ops.full(ref_dy), ops.Act(want, 2, 4 * c))
    got = torch.empty_like(want).fill_(7.0)
    dst = ops.Act(got, 2, 4 * c)
    ops.act_bwd_ex(ops.full(gv), ya, dst, instance_norm=True, unshuffle=True)
    assert torch.equal(got, want)
    if dst.amax is not None:
        assert abs(ops.amax_value(dst.amax) - ref_dy.abs().max().item()) == 0
    acc = g(philox("abx.acc", (n, c, h, w)))
    want2 = acc + ref_dy
    ops.act_bwd_ex(ops.full(gv), ya, ops.full(acc), instance_norm=True, accumulate=True)
    assert torch.equal(acc, want2)


def test_plane_activation_backward_is_bit_stable_beside_another_streams_convolutions(S):
    """[round 4] The one-pass InstanceNorm backward (act_bwd_plane_kernel) on one stream while data-gradient convolutions run on another,
    with no memory in common: every launch gives the bits of the launch that ran alone.  With packed-fp32 instructions in that
    kernel 2 of 3 launches differed on MI355X (16 elements of a plane off by s * yh * (m1 - m2): csrc/san_common.h SAN_NO_PK32,
    scratch/two_stream_probe.py)."""
    ops, Act = S.ops, S.ops.Act
    torch.manual_seed(0)
    aux = torch.cuda.Stream()
    n, c, h, w = 15, 32, 160, 92
    gbuf, y = torch.randn(n, c, h, w, device=DEV), torch.randn(n, c, h, w, device=DEV)
    sc, sh = torch.rand(n, c, device=DEV) + 0.5, torch.randn(n, c, device=DEV) * 0.1
    out = torch.empty_like(gbuf)
    ar_v, ar_a = ops.Arena(), ops.Arena()
    wgt = torch.randn(64, 64, 3, 3, device=DEV) * 0.05
    dy, dx = torch.randn(1, 64, 160, 92, device=DEV), torch.empty(1, 64, 160, 92, device=DEV)

    def victim():
        ops.act_bwd(ops.full(gbuf), Act(y, 0, c, sc, sh, 0.2), ops.full(out), instance_norm=True)

    with ops.use_arena(ar_v):
        victim()
    torch.cuda.synchronize()
    want = out.clone()
    bad = 0
    for _ in range(60):
        with ops.use_arena(ar_a):
            for _ in range(4):
                ops.conv2d_dgrad(ops.full(dy), wgt, ops.full(dx))
        with torch.cuda.stream(aux), ops.use_arena(ar_v):
            victim()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, want))
    assert bad == 0, f"{bad} of 60 launches differ from the launch that ran alone"
