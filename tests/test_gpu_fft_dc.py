"""GPU parity tests (through the C ABI), component: FFT family and the fused cascade boundary (fft2 / ifft2 / rss, sens_reduce / sens_expand, dc_rows).
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


# ---------------------------------------------------------------- FFT family
@pytest.mark.parametrize("shape", [(2, 2, 32, 32), (1, 3, 48, 80), (1, 1, 46, 368), (2, 1, 320, 320), (1, 2, 640, 368),
                                   (1, 1, 30, 45), (1, 1, 7, 11)])
def test_fft2_ifft2(S, shape):
    """[round 1]"""
    x = cplx("hipfft" + str(shape), shape)
    for inv, ref in ((False, S.O.fft2(x)), (True, S.O.ifft2(x))):
        got = S.ops.fft2c(g(x), inverse=inv).cpu()
        assert rel_err(got, ref) < 2e-6, (shape, inv)
    # round trip at full size: ifft2(fft2(x)) == x
    back = S.ops.fft2c(S.ops.fft2c(g(x)), inverse=True).cpu()
    assert rel_err(back, x) < 2e-6


def test_fft_golden(S, ops_golden):
    """[round 1]"""
    for tag, shp in (("32", (2, 2, 32, 32)), ("48x80", (1, 3, 48, 80)), ("46x368", (1, 1, 46, 368))):
        x = cplx("fft." + tag, shp)
        assert rel_err(S.sig.fft2(g(x)).cpu(), as_t(ops_golden[f"fft2_{tag}"], True)) < 2e-6
        assert rel_err(S.sig.ifft2(g(x)).cpu(), as_t(ops_golden[f"ifft2_{tag}"], True)) < 2e-6
        assert rel_err(S.sig.rss(g(x)).cpu(), as_t(ops_golden[f"rss_c_{tag}"])) < 1e-6
        assert rel_err(S.sig.rss(g(x.real.contiguous())).cpu(), as_t(ops_golden[f"rss_r_{tag}"])) < 1e-6


def test_fft_linearity_and_parseval_full_size(S):
    """[round 1] Size-independent properties at BASELINE's full size (N=8, 320x320)."""
    a, b = cplx("lin.a", (8, 1, 320, 320)), cplx("lin.b", (8, 1, 320, 320))
    fa, fb = S.ops.fft2c(g(a)), S.ops.fft2c(g(b))
    fab = S.ops.fft2c(g(a * 2.0 - b * 0.5))
    assert rel_err((fa * 2.0 - fb * 0.5).cpu(), fab.cpu()) < 2e-6
    e_x = (a.abs().double() ** 2).sum().item()
    e_k = (fa.cpu().abs().double() ** 2).sum().item()
    assert abs(e_x - e_k) / e_x < 1e-6      # ortho transform preserves energy


@pytest.mark.parametrize("shape", [(2, 3, 32, 48), (2, 1, 320, 320), (1, 15, 64, 368)])
def test_sens_reduce_expand_dc_rss(S, shape):
    """[round 1]"""
    n, c, h, w = shape
    k, k0, s = cplx("sr.k", shape), cplx("sr.k0", shape), cplx("sr.s", shape)
    r = cplx("sr.r", (n, 1, h, w))
    mask = (philox("sr.m", (w,)) > 0.3)
    dcw = torch.tensor([0.73])
    # sens_reduce -> planar (written into a 3-channel buffer like the cascades do)
    out = torch.zeros((n, 3, h, w), device=DEV)
    S.ops.sens_reduce(g(k), g(s), out)
    want = S.O.sens_reduce(k, s)
    got = torch.complex(out[:, 0:1], out[:, 1:2]).cpu()
    assert rel_err(got, want) < 3e-6
    assert out[:, 2].abs().max().item() == 0.0
    # sens_expand + soft DC + combine
    rp = torch.cat([r.real, r.imag], 1)
    kout = torch.empty_like(g(k))
    S.ops.sens_expand_dc(g(rp), g(s), g(k), g(k0), g(mask.float()), g(dcw), kout)
    zero = torch.zeros(1, 1, 1, 1, dtype=k.dtype)
    want = k - torch.where(mask, k - k0, zero) * dcw - S.O.sens_expand(r, s)
    assert rel_err(kout.cpu(), want) < 3e-6
    # in-place variant (k_out aliases k) gives the same answer
    kk = g(k).clone()
    S.ops.sens_expand_dc(g(rp), g(s), kk, g(k0), g(mask.float()), g(dcw), kk)
    assert torch.equal(kk, kout)
    # rss(ifft2(k))
    assert rel_err(S.ops.ifft2_rss(g(k)).cpu(), S.O.rss(S.O.ifft2(k))) < 3e-6


def test_sens_golden(S, ops_golden):
    """[round 1]"""
    k, s, img = cplx("blk.k", (2, 3, 32, 48)), cplx("blk.s", (2, 3, 32, 48)), cplx("blk.img", (2, 1, 32, 48))
    out = torch.empty((2, 2, 32, 48), device=DEV)
    S.ops.sens_reduce(g(k), g(s), out)
    assert rel_err(torch.complex(out[:, 0:1], out[:, 1:2]).cpu(), as_t(ops_golden["sens_reduce"], True)) < 3e-6
    rp = torch.cat([img.real, img.imag], 1)
    z = torch.zeros_like(g(k))
    kout = torch.empty_like(z)
    S.ops.sens_expand_dc(g(rp), g(s), z, z, g(torch.zeros(48)), g(torch.zeros(1)), kout)
    assert rel_err((-kout).cpu(), as_t(ops_golden["sens_expand"], True)) < 3e-6


@pytest.mark.parametrize("n,c,h,w", [(2, 1, 320, 320), (1, 3, 320, 320), (2, 2, 48, 80), (1, 1, 320, 64)])
def test_fused_cascade_boundary(S, n, c, h, w):
    """[round 1] san_sens_expand_dc_next + san_sens_reduce_from_cols / san_ifft2_rss_from_cols == the unfused calls (the
    320-row case runs the fused register-resident kernel, the others the two-launch fallback).  1e-6 relative."""
    ops = S.ops
    k = torch.complex(philox("fc.kr", (n, c, h, w)), philox("fc.ki", (n, c, h, w)))
    k0 = torch.complex(philox("fc.k0r", (n, c, h, w)), philox("fc.k0i", (n, c, h, w)))
    sens = torch.complex(philox("fc.sr", (n, c, h, w)), philox("fc.si", (n, c, h, w)))
    r = philox("fc.r", (n, 2, h, w))
    mask = (philox("fc.m", (w,)) > 0).float()
    dcw = torch.tensor([0.7])
    kd, k0d, sd, rd, md, dd = g(k), g(k0), g(sens), g(r), g(mask), g(dcw)
    ref_k = ops.sens_expand_dc(rd, sd, kd, k0d, md, dd, torch.empty_like(kd))
    ref_m = ops.sens_reduce(ref_k, sd, torch.empty((n, 2, h, w), device=DEV))
    ref_rss = ops.ifft2_rss(ref_k)
    cols = torch.empty_like(kd)
    got_k = ops.sens_expand_dc(rd, sd, kd, k0d, md, dd, torch.empty_like(kd), next_cols=cols)
    got_m = ops.sens_reduce(got_k, sd, torch.empty((n, 2, h, w), device=DEV), cols=cols)
    got_rss = ops.ifft2_rss(got_k, cols=cols)
    assert torch.equal(torch.view_as_real(got_k), torch.view_as_real(ref_k))
    assert rel_err(got_m.cpu(), ref_m.cpu()) < 1e-6
    assert rel_err(got_rss.cpu(), ref_rss.cpu()) < 1e-6


# ------------------------------------------------------------------ image-domain cascade boundary
@pytest.mark.parametrize("n,c,h,w", [(2, 1, 320, 320), (1, 3, 320, 320), (2, 3, 48, 80), (1, 2, 46, 368), (2, 1, 30, 45), (1, 1, 6, 320),
                                     (1, 2, 7, 320), (2, 1, 9, 368), (1, 15, 8, 368)])
def test_dc_rows_vs_kspace_formula(S, n, c, h, w):
    """[round 2] san_dc_rows (one row-local launch per cascade on x = ifft2(k)) against the reference's k-space update
    k' = k - w where(M, k - k0, 0) - fft2(r S), m' = sum_c ifft2(k')_c conj(S_c) (varnet.py:508-530) evaluated in
    float64 on the CPU; backward form against autograd of the same expression; row lengths 320 (register kernel, two rows
    per wave: odd heights leave a half-empty last wave), 80 / 45 (radix 2-5) and 368 (the 23 x 16 register kernel, four rows
    per wave, with the in-kernel coil combination of a single coil and the separate pass for several)."""
    F = torch.fft
    x = cplx("dcr.x", (n, c, h, w))
    sens = cplx("dcr.s", (n, c, h, w))
    sens = sens / (S.O.rss(sens) + 1e-6)
    k0 = cplx("dcr.k0", (n, c, h, w))
    r = cplx("dcr.r", (n, 1, h, w))
    mask = (philox("dcr.m", (w,)) > 0.3).float()
    mask[:3] = 1
    dcw = torch.tensor([0.8])
    k0 = k0 * mask                                                     # a masked acquisition
    x64 = x.to(torch.complex128).requires_grad_(True)
    s64, r64 = sens.to(torch.complex128).requires_grad_(True), r.to(torch.complex128).requires_grad_(True)
    w64 = dcw.double().requires_grad_(True)
    k = F.fft2(x64, norm="ortho")
    k1 = k - w64 * torch.where(mask.bool(), k - k0.to(torch.complex128), torch.zeros((), dtype=torch.complex128)) - F.fft2(r64 * s64, norm="ortho")
    x1 = F.ifft2(k1, norm="ortho")
    m1 = (x1 * s64.conj()).sum(1, keepdim=True)
    # forward through the library
    k0x = S.ops.fft_cols(g(k0), True)
    assert rel_err(k0x.cpu(), F.ifft(k0.to(torch.complex128), dim=-2, norm="ortho")) < 3e-6
    r_planar = g(torch.cat([r.real, r.imag], 1))
    x_out = torch.empty((n, c, h, w), device=DEV, dtype=torch.complex64)
    m_out = torch.zeros((n, 3, h, w), device=DEV)
    dk = torch.empty_like(x_out)
    S.ops.dc_rows(g(x), g(sens), k0x, g(mask), g(dcw), r_planar, x_out, m_out, dk)
    assert rel_err(x_out.cpu(), x1.detach()) < 3e-6
    assert rel_err(torch.complex(m_out[:, 0:1], m_out[:, 1:2]).cpu(), m1.detach()) < 3e-6
    assert m_out[:, 2].abs().sum().item() == 0                       # channel 2 (the reference image) untouched
    xa = g(x).clone()
    S.ops.dc_rows(xa, g(sens), k0x, g(mask), g(dcw), r_planar, xa, None)     # in place, no coil combination
    assert torch.equal(xa, x_out)
    # backward: L = Re sum conj(gw) x'  ->  dL/dx, dL/dr, dL/dw, and the propagation / sensitivity-map pass
    gw = cplx("dcr.g", (n, c, h, w))
    (x1 * gw.to(torch.complex128).conj()).real.sum().backward()
    g_d = torch.empty_like(x_out)
    g_r = torch.empty((n, 2, h, w), device=DEV)
    d_w = S.ops.dc_rows_bwd(g(gw), g(sens), g(mask), g(dcw), g_d, g_r, dk)
    assert rel_err(g_d.cpu(), x64.grad) < 3e-6                        # (no path through m here: gd only)
    assert rel_err(torch.complex(g_r[:, 0:1], g_r[:, 1:2]).cpu(), r64.grad) < 3e-6
    assert abs(d_w.item() - w64.grad.item()) < 3e-5 * max(1.0, abs(w64.grad.item()))
    # dL/dS of x' (only the -r S term depends on S) and gd += gm S
    gm = cplx("dcr.gm", (n, 1, h, w))
    gS = torch.zeros_like(x_out)
    g_d2 = g_d.clone()
    S.ops.sens_grad_prop(gS, r_planar, g(gw), g(x), g(torch.cat([gm.real, gm.imag], 1)), g_d2, g(sens))
    assert rel_err(g_d2.cpu(), x64.grad + gm.to(torch.complex128) * sens.to(torch.complex128)) < 3e-6
    want_gs = s64.grad + gm.to(torch.complex128).conj() * x.to(torch.complex128)      # + the m = sum conj(S) x term
    assert rel_err(gS.cpu(), want_gs) < 3e-6


@pytest.mark.parametrize("n,c,h,w", [(8, 1, 320, 320), (1, 15, 320, 320), (2, 3, 101, 320), (1, 1, 6, 320)])
def test_cascade_boundary_emits_the_next_normunet_statistics(S, n, c, h, w):
    """[round 6] san_dc_rows_stats == san_dc_rows bit for bit on x' and m', plus (count, mean, M2) records of m's two planes that
    san_norm_finalize merges into the same statistics san_plane_stats' records give (the next cascade's NormUnet.norm,
    varnet.py:262-273): mean / variance within 2e-6 of float64 on the stored planes; two lines per wave, one line per wave
    (15 coils: under one wave per SIMD), an odd height; other row lengths report 0 tiles (the caller keeps san_plane_stats)."""
    ops = S.ops
    assert ops.lib().query("san_dc_rows_stat_tiles", 2, 1, 46, 368) == 0 and ops.lib().query("san_dc_rows_stat_tiles", 2, 3, 48, 80) == 0
    x, sens, k0 = g(cplx("dcs.x", (n, c, h, w))), g(cplx("dcs.s", (n, c, h, w))), g(cplx("dcs.k0", (n, c, h, w)))
    r = g(philox("dcs.r", (n, 2, h, w)))
    mask = g((philox("dcs.m", (w,)) > 0.3).float())
    dcw = g(torch.tensor([0.7]))
    k0x = k0                                     # (any data term: the launch is row-local)
    outs = []
    for with_stats in (False, True):
        x_out = torch.empty_like(x)
        m_out = torch.zeros((n, 3, h, w), device=DEV)
        st = ops.dc_rows_stat_part(n, c, h, w, DEV, tag=".t") if with_stats else None
        if with_stats:
            assert st is not None
            st.fill_(float("nan"))
        ops.dc_rows(x, sens, k0x, mask, dcw, r, x_out, m_out, None, m_stats=st)
        torch.cuda.synchronize()
        outs.append((x_out, m_out, st))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    m_out, st = outs[1][1], outs[1][2]
    assert bool(torch.isfinite(st).all()) and float(st[..., 0].sum(dim=2).min()) == h * w == float(st[..., 0].sum(dim=2).max())
    m64 = m_out[:, :2].double()
    var, mean = torch.var_mean(m64, dim=(2, 3), unbiased=False)
    for part in (st, ops.plane_stats(ops.Act(m_out, 0, 2), tag=".dcs")):
        sc, sh = torch.zeros((n, 2), device=DEV), torch.zeros((n, 2), device=DEV)
        ops.norm_finalize(part, ops.NORM_INSTANCE, 0.0, sc, sh, 0)
        torch.cuda.synchronize()
        assert ((sc.double() - torch.rsqrt(var)).abs() / torch.rsqrt(var)).max().item() < 2e-6
        assert ((sh.double() + mean * torch.rsqrt(var)).abs().max() / (mean * torch.rsqrt(var)).abs().max().clamp_min(1e-3)).item() < 1e-5


# ------------------------------------------------------------------ single layers
def test_varnetblock_step_and_sens_expand_golden(S, ops_golden):
    """[round 2] One cascade with a real regulariser through VarNetBlock.forward, and the stand-alone sens_expand
    (varnet.py:508-530), against the reference."""
    gold = load_golden("layers_small.npz")
    blk = S.varnet.VarNetBlock(S.varnet.NormUnet(4, 2, use_ref=True))
    _load(S, blk, 21)
    blk.to(DEV)
    k, k0, sens = cplx("vb.k", (2, 3, 32, 48)), cplx("vb.k0", (2, 3, 32, 48)), cplx("vb.s", (2, 3, 32, 48))
    sens = sens / (S.O.rss(sens) + 1e-6)
    ref = philox("vb.ref", (2, 1, 32, 48), lo=0.0, hi=1.0)
    mask = torch.from_numpy(gold["varnetblock.mask"])
    with torch.no_grad():
        got = blk(g(k), g(k0), mask.to(DEV), g(sens), g(ref))
    assert rel_err(got.cpu(), as_t(gold["varnetblock"], True)) < 2e-5
    blk0 = S.varnet.VarNetBlock(torch.nn.Identity()).to(DEV)
    img, s = cplx("blk.img", (2, 1, 32, 48)), cplx("blk.s", (2, 3, 32, 48))
    out = blk0.sens_expand(g(img), g(s))
    assert rel_err(out.cpu(), as_t(ops_golden["sens_expand"], True)) < 3e-6
    red = blk0.sens_reduce(g(cplx("blk.k", (2, 3, 32, 48))), g(s))
    assert rel_err(red.cpu(), as_t(ops_golden["sens_reduce"], True)) < 3e-6
