"""GPU parity tests (through the C ABI), component: end-to-end: VarNet / NormUnet / alignment net against the golden fixtures, training steps, narrow-precision PSNR (rows a6, a7, a10, a16, a17).
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


@pytest.mark.parametrize("tag,shape", [("32", (2, 1, 32, 32)), ("48x80c3", (2, 3, 48, 80))])
def test_e2e_small_golden(S, tag, shape):
    """[round 1]"""
    gold = load_golden(f"e2e_small_{tag}.npz")
    n, c, h, w = shape
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    net_T, net_R = _build_nets(S, gold, c, 2, 4, 2, 2)
    net_T.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net_T.state_dict().items()], seed=41))
    net_R.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net_R.state_dict().items()], seed=42))
    net_T.to(DEV).eval()
    net_R.to(DEV).eval()
    o = _run_pipeline(S, net_T, net_R, img_full, img_aux, pruned, w, 0.25)
    assert rel_err(o["img_k_sampled"].cpu(), as_t(gold["eval.img_k_sampled"], True)) < 3e-6
    assert rel_err(o["img_sampled"].cpu(), as_t(gold["eval.img_sampled"], True)) < 3e-6
    assert rel_err(o["img_offset"].cpu(), as_t(gold["eval.img_offset"])) < 3e-5
    assert rel_err(o["img_warped"].cpu(), as_t(gold["eval.img_warped"])) < 3e-5
    assert rel_err(o["img_rec"].cpu(), as_t(gold["eval.img_rec"])) < 1e-4          # north_star bar
    assert abs(o["loss_sim"].item() - float(gold["eval.loss_sim"])) < 2e-5
    ls = float(gold["eval.loss_smooth"])
    assert abs(o["loss_smooth"].item() - ls) < 1e-4 * max(abs(ls), 1e-6)


def test_e2e_full_320_golden(S):
    """[round 1] Config-2 network (12 cascades, chans 18, sens_chans 8) at 320x320, N=1,
    against the reference's own fp32 output, with its fp64 run as arbiter."""
    gold = load_golden("e2e_full_320.npz")
    n, c, h, w = 1, 1, 320, 320
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=1234)
    assert torch.equal(img_full.real, as_t(gold["img_full_re"]))
    pruned = as_t(gold["pruned"])
    net_T = S.cross.SpatialTransformer(1)
    net_R = S.varnet.VarNet(num_cascades=12, sens_chans=8, sens_pools=4, chans=18, pools=4, use_ref=True)
    net_T.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net_T.state_dict().items()], seed=1235))
    net_R.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net_R.state_dict().items()], seed=1236))
    net_T.to(DEV).eval()
    net_R.to(DEV).eval()
    o = _run_pipeline(S, net_T, net_R, img_full, img_aux, pruned, w, 0.25)
    ref32, ref64 = as_t(gold["img_rec"]), as_t(gold["img_rec_f64"])
    e_ref = rel_err(ref32, ref64)                      # the reference's own fp32 noise (about 4.5e-5)
    e32 = rel_err(o["img_rec"].cpu(), ref32)
    e64 = rel_err(o["img_rec"].cpu(), ref64)
    print(f"rec rel-L2: hip-vs-ref32 {e32:.2e}, hip-vs-ref64 {e64:.2e}, ref32-vs-ref64 {e_ref:.2e}")
    assert rel_err(o["img_offset"].cpu(), as_t(gold["img_offset"])) < 3e-5
    assert rel_err(o["img_warped"].cpu(), as_t(gold["img_warped"])) < 3e-5
    assert e32 < 1e-4                                   # north_star: within 1e-4 of the CPU reference
    assert e64 < max(1e-4, 2 * e_ref)                   # and no further from the truth than the reference is
    assert abs(o["loss_sim"].item() - float(gold["loss_sim"])) < 2e-5
    psnr = S.O.psnr(ref32, o["img_rec"].cpu())
    print(f"PSNR(hip, ref32) = {psnr:.1f} dB")
    assert psnr > 80.0


def test_unet_backward_vs_oracle_autograd(S):
    """[round 1] Unet.run_bwd (hand-written backward on the HIP kernels) against autograd of the oracle."""
    n, h, w = 2, 32, 48
    net = S.varnet.Unet(3, 2, chans=4, num_pool_layers=2)
    params = S.synth.fill_params([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=77)
    net.load_state_dict(params)
    net.to(DEV)
    x = philox("ub.x", (n, 3, h, w))
    gout = philox("ub.g", (n, 2, h, w))
    p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
    x64 = x.double().requires_grad_(True)
    S.O.unet_forward(p64, "", x64, 2).backward(gout.double())
    y = torch.empty((n, 2, h, w), device=DEV)
    net.run(S.ops.full(g(x)), S.ops.full(y), key="t")
    assert rel_err(y.cpu(), S.O.unet_forward(params, "", x, 2)) < 1e-5
    gx = net.run_bwd(g(gout), key="t")
    assert rel_err(gx.cpu(), x64.grad.float()) < 1e-4
    worst = 0.0
    for name, prm in net.named_parameters():
        want = p64[name].grad.float()
        err = (prm.grad.cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-12)
        worst = max(worst, err)
    assert worst < 2e-4, worst


@pytest.mark.parametrize("use_ref", [True, False])
def test_normunet_backward_vs_oracle_autograd(S, use_ref):
    """[round 1]"""
    n, h, w = 2, 32, 48
    net = S.varnet.NormUnet(4, 2, use_ref=use_ref)
    params = S.synth.fill_params([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=78)
    net.load_state_dict(params)
    net.to(DEV)
    x = cplx("nb.x", (n, 1, h, w)) * 2 + 0.5
    ref = philox("nb.ref", (n, 1, h, w), lo=0.0, hi=1.0) if use_ref else None
    gout = cplx("nb.g", (n, 1, h, w))
    p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
    x64 = x.to(torch.complex128).requires_grad_(True)
    r64 = ref.double().requires_grad_(True) if use_ref else None
    y64 = S.O.normunet_forward(p64, "", x64, r64, 2, use_ref)
    (y64.real * gout.real.double() + y64.imag * gout.imag.double()).sum().backward()
    # HIP path through the fused entry points
    xin = net.input_buffer(n, h, w, DEV, "nbt")
    planar = torch.cat([x.real, x.imag], 1)
    S.ops.apply(S.ops.full(g(planar)), xin.view(0, 2))
    if use_ref:
        net.set_ref(xin, g(ref))
    out = torch.empty((n, 2, h, w), device=DEV)
    net.run(xin, out, "nbt")
    assert rel_err(torch.complex(out[:, 0:1], out[:, 1:2]).cpu(), y64.detach().to(torch.complex64)) < 2e-5
    g_planar = torch.cat([gout.real, gout.imag], 1)
    g_m, g_ref = net.run_bwd(g(g_planar), "nbt", want_ref_grad=use_ref)
    want = torch.cat([x64.grad.real, x64.grad.imag], 1).float()
    assert rel_err(g_m.cpu(), want) < 2e-4
    if use_ref:
        assert rel_err(g_ref.cpu(), r64.grad.float()) < 2e-4
    worst = 0.0
    for name, prm in net.named_parameters():
        wantp = p64[name].grad.float()
        worst = max(worst, (prm.grad.cpu() - wantp).abs().max().item() / max(wantp.abs().max().item(), 1e-12))
    assert worst < 3e-4, worst


@pytest.mark.parametrize("tag,shape", [("32", (2, 1, 32, 32)), ("48x80c3", (2, 3, 48, 80))])
def test_varnet_backward_vs_golden_grads(S, tag, shape):
    """[round 1] VarNet.backward (all cascades, dc weights, sensitivity net) against the gradients the
    REFERENCE produced for the same weights and inputs (tests/golden, 'Rec' objective)."""
    gold = load_golden(f"e2e_small_{tag}.npz")
    n, c, h, w = shape
    net_R = S.varnet.VarNet(num_cascades=2, sens_chans=2, sens_pools=2, chans=4, pools=2, use_ref=True)
    net_R.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net_R.state_dict().items()], seed=42))
    net_R.to(DEV).train()
    pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    k_samp = g(as_t(gold["train.img_k_sampled"], True))
    warped = g(as_t(gold["train.img_warped"]))           # teacher-forced: the reference's own train-mode warp
    full_rss = g(as_t(gold["train.img_full_rss"]))
    rec = net_R(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32))
    assert rel_err(rec.cpu(), as_t(gold["train.img_rec"])) < 1e-4
    g_img = S.ops.ssim_loss_bwd(full_rss, rec, 1.0)      # weight_sim = 1
    net_R.backward(g_img, want_ref_grad=False)
    worst, worst_name = 0.0, ""
    for name, prm in net_R.named_parameters():
        want = as_t(gold["grad.R." + name])
        scale = want.abs().max().item()
        if scale < 1e-12:
            continue
        err = (prm.grad.cpu() - want).abs().max().item() / scale
        if err > worst:
            worst, worst_name = err, name
    print("worst relative gradient error", worst, worst_name)
    assert worst < 2e-4, (worst, worst_name)           # measured 2e-5 (bf16x3 kernels: layers here have < 16 channels)


@pytest.mark.parametrize("tag,shape", [("32", (2, 1, 32, 32)), ("48x80c3", (2, 3, 48, 80))])
def test_full_rec_step_gradients_vs_golden(S, fp32_convs, tag, shape):
    """[round 1] CSModel-style 'Rec' step: train-mode forward of T and R, hand-written backward through SSIM,
    VarNet, warp and the BatchNorm alignment network; losses, BatchNorm running statistics and EVERY
    parameter gradient against what the reference produced (tests/golden)."""
    from spatialalignmentnetwork_amd.basemodel import Config
    from spatialalignmentnetwork_amd.model import CSModel
    gold = load_golden(f"e2e_small_{tag}.npz")
    n, c, h, w = shape
    cfg = Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                 weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=4,
                 sens_chans=2, pools=2, sens_pools=2)
    net = CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    net.net_T.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net.net_T.state_dict().items()], seed=41))
    net.net_R.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net.net_R.state_dict().items()], seed=42))
    net.to(DEV).train()
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    net.set_input(g(img_full), g(img_aux))
    net.loss_all = 0
    net.forwardT()
    net.forwardR()
    assert rel_err(net.img_warped.cpu(), as_t(gold["train.img_warped"])) < 5e-5
    assert rel_err(net.img_rec.cpu(), as_t(gold["train.img_rec"])) < 1e-4
    assert abs(net.loss_all.item() - float(gold["train.loss_all"])) < 1e-4 * max(1.0, abs(float(gold["train.loss_all"])))
    net.backward(train_T=True)
    for pre, mod in (("grad.R.", net.net_R), ("grad.T.", net.net_T)):
        worst, worst_name = 0.0, ""
        for name, prm in mod.named_parameters():
            want = as_t(gold[pre + name])
            scale = want.abs().max().item()
            got = prm.grad.cpu() if prm.grad is not None else torch.zeros_like(want)
            if scale < 1e-7:
                assert got.abs().max().item() < 1e-5, name      # conv bias in front of BatchNorm: exactly 0 in theory
                continue
            err = (got - want).abs().max().item() / scale
            if err > worst:
                worst, worst_name = err, name
        print(pre, "worst relative gradient error", worst, worst_name)
        # measured (fp32 conv kernels): 2e-5 on both networks at 32 x 32; at 48 x 80 / 3 coils 2e-5 on net_R and 8e-4 on
        # one BatchNorm bias of net_T (a 3 x 5 pixel layer next to a LeakyReLU kink).  Bars = 10x measured.
        bar = 8e-3 if (tag == "48x80c3" and pre == "grad.T.") else 2e-4
        assert worst < bar, (pre, worst, worst_name)
    for k in gold.files:
        if k.startswith("bn_after.T."):
            got = dict(net.net_T.named_buffers())[k[len("bn_after.T."):]]
            assert torch.allclose(got.cpu(), as_t(gold[k]), rtol=2e-4, atol=2e-6), k


def test_full_rec_step_with_bf16x3_convs(S):
    """[round 1] The same 'Rec' step (48 x 80, 3 coils) with the bf16x3 convolution kernels switched on for the layers they
    take: forward outputs and losses at the same bars as the fp32 path; gradients compared NORM-WISE per network
    (relative L2 over all parameters; see the fp32_convs fixture for why element-wise is not meaningful): 5e-3."""
    from spatialalignmentnetwork_amd.basemodel import Config
    from spatialalignmentnetwork_amd.model import CSModel
    assert S.ops.USE_BF16X3[0]
    gold = load_golden("e2e_small_48x80c3.npz")
    n, c, h, w = 2, 3, 48, 80
    cfg = Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                 weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=4,
                 sens_chans=2, pools=2, sens_pools=2)
    net = CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    net.net_T.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net.net_T.state_dict().items()], seed=41))
    net.net_R.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net.net_R.state_dict().items()], seed=42))
    net.to(DEV).train()
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    net.set_input(g(img_full), g(img_aux))
    net.loss_all = 0
    net.forwardT()
    net.forwardR()
    assert rel_err(net.img_warped.cpu(), as_t(gold["train.img_warped"])) < 5e-5
    assert rel_err(net.img_rec.cpu(), as_t(gold["train.img_rec"])) < 1e-4
    assert abs(net.loss_all.item() - float(gold["train.loss_all"])) < 1e-4 * max(1.0, abs(float(gold["train.loss_all"])))
    net.backward(train_T=True)
    for pre, mod in (("grad.R.", net.net_R), ("grad.T.", net.net_T)):
        num = den = 0.0
        for name, prm in mod.named_parameters():
            want = as_t(gold[pre + name]).double()
            got = prm.grad.cpu().double() if prm.grad is not None else torch.zeros_like(want)
            num += ((got - want) ** 2).sum().item()
            den += (want ** 2).sum().item()
        print(pre, "relative L2 over all parameter gradients", (num / den) ** 0.5)
        # measured 8.6e-6 (net_R) / 6.2e-4 (net_T: the BatchNorm layers at 3 x 5 pixels); bars = 10x measured
        assert (num / den) ** 0.5 < (1e-4 if pre == "grad.R." else 6e-3)


def test_normunet_pad_golden(S):
    """[round 2] NormUnet at 50 x 70 (zero pad of the normalised image to 64 x 80, crop, un-normalise; varnet.py:275-332) forward
    and hand-written backward against the reference's output and autograd gradients.  Measured 2e-6 / 3e-5."""
    gold = load_golden("pad_small.npz")
    n, h, w = 2, 50, 70
    net = S.varnet.NormUnet(4, 2, use_ref=True)
    _load(S, net, 51)
    net.to(DEV)
    x, ref, wgt = cplx("pad.x", (n, 1, h, w)), philox("pad.ref", (n, 1, h, w), lo=0.0, hi=1.0), cplx("pad.w", (n, 1, h, w))
    y = net(g(x), g(ref))                                                           # the reference-compatible entry
    assert rel_err(y.cpu(), as_t(gold["nu.y"], True)) < 2e-5
    xin = net.input_buffer(n, h, w, DEV, "padt")
    S.ops.apply(S.ops.full(g(torch.cat([x.real, x.imag], 1))), xin.view(0, 2))
    net.set_ref(xin, g(ref))
    out = torch.empty((n, 2, h, w), device=DEV)
    net.run(xin, out, "padt")
    assert rel_err(torch.complex(out[:, 0:1], out[:, 1:2]).cpu(), as_t(gold["nu.y"], True)) < 2e-5
    g_m, g_ref = net.run_bwd(g(torch.cat([wgt.real, wgt.imag], 1)), "padt", want_ref_grad=True)
    want = as_t(gold["nu.gx"], True)
    # d/dx of Re sum(y conj(w)) under torch's convention for complex leaves: grad = dL/dRe + i dL/dIm
    assert rel_err(torch.complex(g_m[:, 0:1], g_m[:, 1:2]).cpu(), want) < 2e-4
    assert rel_err(g_ref.cpu(), as_t(gold["nu.gref"])) < 2e-4
    worst = 0.0
    for name, prm in net.named_parameters():
        wantp = as_t(gold["nu.grad." + name])
        worst = max(worst, (prm.grad.cpu() - wantp).abs().max().item() / max(wantp.abs().max().item(), 1e-12))
    print("NormUnet 50x70 worst relative parameter-gradient error", worst)
    assert worst < 5e-4, worst


def test_unet_reflect_pad_golden(S):
    """[round 2] Bare U-Net at 25 x 35 with 2 pooling levels: the avg-pool drops odd rows / columns and the up path reflect-pads
    (varnet.py:99,107-114).  Forward and backward against the reference."""
    gold = load_golden("pad_small.npz")
    net = S.varnet.Unet(3, 2, chans=4, num_pool_layers=2)
    _load(S, net, 52)
    net.to(DEV)
    x, gw = philox("pad.u", (2, 3, 25, 35)), philox("pad.uw", (2, 2, 25, 35))
    y = net(g(x))
    assert rel_err(y.cpu(), as_t(gold["un.y"])) < 1e-5
    gx = net.run_bwd(g(gw), key="unet")
    assert rel_err(gx.cpu(), as_t(gold["un.gx"])) < 1e-4
    worst = 0.0
    for name, prm in net.named_parameters():
        want = as_t(gold["un.grad." + name])
        worst = max(worst, (prm.grad.cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-12))
    print("U-Net 25x35 worst relative parameter-gradient error", worst)
    assert worst < 5e-4, worst


def test_varnet_pad_golden(S):
    """[round 2] VarNet (2 cascades, 2 coils, sensitivity net) on 50 x 70 slices, eval."""
    gold = load_golden("pad_small.npz")
    net = S.varnet.VarNet(num_cascades=2, sens_chans=2, sens_pools=2, chans=4, pools=2, use_ref=True)
    _load(S, net, 53)
    net.to(DEV).eval()
    img, _ = S.synth.phantom_pair(2, 2, 50, 70, seed=54)
    pruned = S.synth.equispaced_pruned(70, 0.25, 0)
    refv = philox("pad.vref", (2, 2, 50, 70), lo=0.0, hi=1.0)
    with torch.no_grad():
        ks = S.ops.fft2c(g(img), colmask_out=(~pruned).float().to(DEV))
        rec = net(ks, (~pruned).to(DEV), g(refv), int(70 * 0.25 * 0.32))
    assert rel_err(rec.cpu(), as_t(gold["vn.rec"])) < 1e-4


@pytest.mark.parametrize("tag,cin,cout,shp", [("conv2d", 6, 16, (2, 6, 24, 40)), ("up", 16, 24, (2, 16, 12, 20)),
                                              ("down", 24, 16, (2, 24, 24, 40))])
def test_alignment_layers_golden(S, tag, cin, cout, shp):
    """[round 2] The alignment backbone's Conv2d / Up / Down factories (unet.py:119-140) in eval and train mode, incl. the
    BatchNorm running statistics after one train-mode call, against the reference."""
    gold = load_golden("layers_small.npz")
    U = S.unet
    seq = {"conv2d": U.Conv2d, "up": U.Up, "down": U.Down}[tag](cin, cout)
    _load(S, seq, 31)
    seq.to(DEV)
    host = U.UNet(2, 4, (4, 4))                 # any instance: only its executor methods are used
    x = philox("stl." + tag, shp)
    n, _, h, w = shp
    for mode in ("eval", "train"):
        seq.train(mode == "train")
        src = S.ops.full(g(x))
        if tag == "down":
            pooled = S.ops.full(torch.empty((n, cin, h // 2, w // 2), device=DEV))
            S.ops.avgpool2(src, pooled)
            raw = U._arena_act(f"t.{tag}.{mode}", n, cout, h // 2, w // 2, DEV)
            host._cba(seq, 1, pooled, raw, "t." + tag)
            y = torch.empty((n, cout, h // 2, w // 2), device=DEV)
            S.ops.apply(raw, S.ops.full(y))
        elif tag == "up":
            raw = U._arena_act(f"t.{tag}.{mode}", n, cout, h, w, DEV)
            host._cba(seq, 1, src, raw, "t." + tag, count_scale=4)
            y = torch.empty((n, cout, 2 * h, 2 * w), device=DEV)
            S.ops.upsample2(raw, S.ops.full(y))
        else:
            raw = U._arena_act(f"t.{tag}.{mode}", n, cout, h, w, DEV)
            host._cba(seq, 0, src, raw, "t." + tag)
            y = torch.empty((n, cout, h, w), device=DEV)
            S.ops.apply(raw, S.ops.full(y))
        assert rel_err(y.cpu(), as_t(gold[f"st.{tag}.{mode}"])) < 5e-6, mode
    bn = [m for m in seq if isinstance(m, torch.nn.BatchNorm2d)][0]
    assert torch.allclose(bn.running_mean.cpu(), as_t(gold[f"st.{tag}.running_mean"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu(), as_t(gold[f"st.{tag}.running_var"]), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ CSModel protocol
def test_csmodel_protocol_scalars_save_load(S, tmp_path):
    """[round 2] set_input -> test() -> get_vis() -> save() -> load() on the GPU (model.py:89-121,265-321; basemodel.py:159-182)
    against the scalars and images the REFERENCE's CSModel produced on CPU (tests/golden/csmodel_scalars.npz; the
    reference's hard-coded 8-cascade VarNet at 64 x 64, N = 2).  loss_gan_sim belongs to the GAN branch (out of scope)."""
    gold = load_golden("csmodel_scalars.npz")
    shape = 64
    cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=shape, coils=1, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                        weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False)
    net = S.model.CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(shape, 0.25, 0)
    _load(S, net.net_T, 61)
    _load(S, net.net_R, 62)
    assert net.to(DEV) is net
    net.eval()
    img_full, img_aux = S.synth.phantom_pair(2, 1, shape, shape, seed=63)
    net.set_input(g(img_full), g(img_aux))
    ret = net.test()
    vis = net.get_vis()
    sc = vis["scalars"]
    want = {k[len("scalar."):]: float(gold[k]) for k in gold.files if k.startswith("scalar.")}
    assert set(want) - set(sc) == {"loss_gan_sim"}, (sorted(want), sorted(sc))
    assert set(sc) <= set(want)
    # measured differences are listed in DESIGN.md section 4; bars are 10x those
    # measured: loss_all / loss_sim 6e-8, loss_smooth 1e-7 relative, MI 1e-9, PSNR 2e-6 dB, SSIM 3e-8, MAE 7e-8, MSE 4e-8
    tol = {"loss_all": 1e-6, "loss_sim": 1e-6, "loss_smooth": 1e-5 * abs(want["loss_smooth"]) + 1e-15, "metric_MI": 1e-4,
           "metric_PSNR": 5e-5, "metric_SSIM": 1e-6, "metric_MAE": 1e-6, "metric_MSE": 1e-6}
    for k_, v in sc.items():
        print(f"{k_}: hip {v:.9g} reference {want[k_]:.9g}")
        assert abs(v - want[k_]) <= tol[k_], (k_, v, want[k_])
    assert ret == -sc["metric_PSNR"] and abs(ret - float(gold["return"])) <= tol["metric_PSNR"]
    for k_ in ("img_full_rss", "img_sampled_rss", "img_aux_rss", "img_warped_rss", "img_rec", "img_mask"):
        assert k_ in vis["images"], k_
        assert rel_err(getattr(net, k_).cpu(), as_t(gold[k_])) < (1e-4 if k_ == "img_rec" else 3e-5), k_
    assert rel_err(net.img_offset.cpu(), as_t(gold["img_offset"])) < 3e-5            # eval.py:70 reads it (NHWC)
    assert "img_offset" not in vis["images"] and torch.equal(vis["histograms"]["weights"]["values"].cpu(), torch.ones(shape))
    # a second set_input must reset every loss_* / img_* / metric_* attribute (model.py:91-98)
    net.set_input(g(img_full), g(img_aux))
    assert not any(k_.startswith(("loss_", "metric_")) for k_ in net.__dict__)
    # checkpoint round trip through the reference's directory format
    ck = str(tmp_path / "ckpt.pt")
    net.save(ck)
    assert sorted(os.listdir(ck)) == ["config", "net_R", "net_T", "net_mask"]
    net2 = S.model.CSModel(ckpt=ck)
    net2.to(DEV).eval()
    net2.set_input(g(img_full), g(img_aux))
    assert net2.test() == ret
    assert torch.equal(net2.img_rec, vis["images"]["img_rec"])


@pytest.mark.parametrize("tag,damp", [("raw", 1.0), ("damped", 0.1)])
def test_train_step_full_320_golden(S, tag, damp):
    """[round 2] One 'Rec' training step at the bench shape (N = 2, 320 x 320, 12 cascades, chans 18; model.py:206-216) with the
    DEFAULT kernel mix (every bf16x3 kernel, weight gradients on the side stream) against the reference's fp32 step,
    with the reference's own fp64 step as arbiter.  Two weight sets (tests/golden/make_golden.py make_train_full):
    'raw' = the default random weights, whose 12 O(1) cascade maps amplify rounding noise (the reference's own fp32 and
    fp64 runs differ by 1.4e-3 on the image and 16 % / 4.6 % on the gradients of net_R / net_T); 'damped' = cascade
    output convolutions x 0.1 (6.5e-5; 1.8 % / 0.7 %).  Bars are in units of those reference-vs-reference distances,
    measured with the same estimators (stored in / recomputed from the fixture)."""
    gold = load_golden("train_full_320.npz")
    assert S.ops.USE_BF16X3[0] and S.ops.WGRAD_OVERLAP[0]
    n, c, h, w = 2, 1, 320, 320
    cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                        weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=12)
    net = S.model.CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    _load(S, net.net_T, 2235)
    net.net_R.load_state_dict(S.synth.fill_params(_shapes(net.net_R), seed=2236, damp=damp))
    net.to(DEV).train()
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=2234)
    net.set_input(g(img_full), g(img_aux))
    net.loss_all = 0
    net.forwardT()
    net.forwardR()
    e_ref = float(gold[f"{tag}.ref32_vs_ref64.img_rec"])
    e32 = rel_err(net.img_rec.cpu(), as_t(gold[f"{tag}.f32.img_rec"]))
    e64 = rel_err(net.img_rec.cpu().double(), as_t(gold[f"{tag}.f64.img_rec"]))
    print(f"[{tag}] train-mode rec: hip-vs-ref32 {e32:.2e}, hip-vs-ref64 {e64:.2e}, ref32-vs-ref64 {e_ref:.2e}")
    assert rel_err(net.img_warped.cpu(), as_t(gold["f32.img_warped"])) < 3e-5
    assert e64 < max(1e-4, 2 * e_ref)                   # no further from the truth than twice the reference itself
    assert e32 < max(1e-4, 3 * e_ref)                   # two fp32 runs of a map with that noise: ~sqrt(2) e_ref expected
    for k_ in ("loss_sim", "loss_smooth", "loss_all"):
        want = float(gold[f"{tag}.f32.{k_}"])
        got = (net.loss_all if k_ == "loss_all" else getattr(net, k_)).item()
        assert abs(got - want) < max(2e-5, 2 * e_ref) * max(1.0, abs(want)), (k_, got, want)
    for o in (net.optim_R, net.optim_T):
        o.zero_grad()
    with S.ops.wgrad_overlap():
        net.backward(train_T=True)
    torch.cuda.synchronize()
    for nt, mod in (("R", net.net_R), ("T", net.net_T)):
        named = [(nm, p.grad) for nm, p in mod.named_parameters()]
        floor = float(gold[f"{tag}.ref32_vs_ref64.grad.{nt}"])                # the reference's own fp32 noise, norm-wise
        d32, d64 = gold[f"{tag}.f32.grad.{nt}.probes"], gold[f"{tag}.f64.grad.{nt}.probes"]
        ne = gold[f"{tag}.f64.grad.{nt}.numel"][:, None]
        floor_probe = float((((d32 - d64) ** 2 * ne / 16.0).sum() / (gold[f"{tag}.f64.grad.{nt}.l2"] ** 2).sum()) ** 0.5)
        l2a, l2b = gold[f"{tag}.f32.grad.{nt}.l2"], gold[f"{tag}.f64.grad.{nt}.l2"]
        big = l2b > 1e-3 * l2b.max()
        floor_norm = float((np.abs(l2a - l2b) / np.maximum(l2b, 1e-30))[big].max())
        wn32, name32, pe32 = _digest_errors_r2(S, named, gold, f"{tag}.f32.grad.{nt}.")
        wn64, name64, pe64 = _digest_errors_r2(S, named, gold, f"{tag}.f64.grad.{nt}.")
        print(f"[{tag}] net_{nt}: per-tensor norm error vs ref32 {wn32:.2e} ({name32}), vs ref64 {wn64:.2e} ({name64}); "
              f"probe-estimated relative L2 vs ref32 {pe32:.2e}, vs ref64 {pe64:.2e}; reference fp32-vs-fp64: exact "
              f"{floor:.2e}, probe-estimated {floor_probe:.2e}, worst per-tensor norm {floor_norm:.2e}")
        # no further from the fp64 truth than 3x the reference's own fp32 run, measured with the same estimators
        assert pe64 < max(3.0 * max(floor, floor_probe), 5e-4), (nt, pe64, floor, floor_probe)
        assert wn64 < max(3.0 * max(floor, floor_norm), 5e-4), (nt, wn64, name64, floor, floor_norm)
    if tag == "raw":
        for k_ in gold.files:
            if k_.startswith("f32.bn_after.T."):
                got = dict(net.net_T.named_buffers())[k_[len("f32.bn_after.T."):]]
                assert torch.allclose(got.cpu(), as_t(gold[k_]), rtol=2e-4, atol=2e-6), k_


def test_cascade_checksums_full_320(S):
    """[round 2] Per-cascade k-space checksums (sum re, sum im, L2) of the 12-cascade network at 320 x 320 against the
    reference's (e2e_full_320.npz: forward hooks on its cascades).  Train-mode forward keeps every cascade's output."""
    gold = load_golden("e2e_full_320.npz")
    w = 320
    img_full, img_aux = S.synth.phantom_pair(1, 1, w, w, seed=1234)
    pruned = as_t(gold["pruned"])
    net_R = S.varnet.VarNet(num_cascades=12, sens_chans=8, sens_pools=4, chans=18, pools=4, use_ref=True)
    _load(S, net_R, 1236)
    net_R.to(DEV).train()                                  # no BatchNorm in VarNet: train == eval arithmetic
    keep = (~pruned).float().to(DEV)
    k_samp = S.ops.fft2c(g(img_full), colmask_out=keep)
    net_R(k_samp, (~pruned).to(DEV), g(as_t(gold["img_warped"])), int(w * 0.25 * 0.32))
    want = gold["cascade_checksums"]
    for j in range(12):
        # the cascades keep the image-domain state x_j = ifft2(k_j): transform it back for the k-space checksums
        xj = S.ops.owner_arena(net_R).get(f"cas{j}.xout", (1, 1, w, w), torch.device(DEV), dtype=torch.complex64)
        k = S.ops.fft2c(xj).cpu()
        got = np.array([k.real.double().sum().item(), k.imag.double().sum().item(), k.abs().double().pow(2).sum().sqrt().item()])
        l2 = want[j, 2]
        # sums of 102,400 values of magnitude ~L2/320 carry ~1e-5 of relative noise through 12 cascades; L2 itself ~1e-5
        assert abs(got[2] - l2) < 1e-5 * l2, (j, got, want[j])            # measured <= 3e-7
        print(j, got - want[j], l2)
        assert abs(got[0] - want[j, 0]) < 2e-4 * l2 and abs(got[1] - want[j, 1]) < 2e-4 * l2, (j, got, want[j])   # measured <= 1.2e-5 l2


def test_e2e_multicoil_640x368_golden(S):
    """[round 2] BASELINE config 4: one 640 x 368 slice, 15 coils, 8x equispaced mask (46 kept columns, 14 low frequencies),
    sensitivity-map VarNet with 12 cascades + the 30-channel alignment network, against the reference's fp32 output
    with its fp64 run as arbiter (varnet.py:389-420,465-486).  FFT length 368 = 2^4 * 23."""
    gold = load_golden("multicoil_640x368.npz")
    n, c, h, w, sp = 1, 15, 640, 368, 0.125
    net_T, net_R = _multicoil_nets(S, 12, 3234)
    net_T.eval()
    net_R.train()                                         # keeps the per-cascade k-space for the checksums below
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=3234)
    pruned = S.synth.equispaced_pruned(w, sp, 0)
    assert torch.equal(pruned, as_t(gold["pruned"])) and int((~pruned).sum()) == 46
    keep = (~pruned).float().to(DEV)
    k_samp = S.ops.fft2c(g(img_full), colmask_out=keep)
    samp = S.sig.ifft2(k_samp)
    aux_abs, samp_abs = S.ops.cabs(g(img_aux)), S.ops.cabs(samp)
    with torch.no_grad():
        offset, grid = net_T(aux_abs, samp_abs)
        warped = net_T.warp(aux_abs, grid)
    rec = net_R(k_samp, (~pruned).to(DEV), warped, int(w * sp * 0.32))
    loss_sim = S.ssim.ssimloss(S.sig.rss(g(img_full)), rec)
    ref32, ref64 = as_t(gold["img_rec"]), as_t(gold["img_rec_f64"])
    e_ref, e32, e64 = rel_err(ref32, ref64), rel_err(rec.cpu(), ref32), rel_err(rec.cpu().double(), ref64)
    print(f"multi-coil rec rel-L2: hip-vs-ref32 {e32:.2e}, hip-vs-ref64 {e64:.2e}, ref32-vs-ref64 {e_ref:.2e}")
    assert rel_err(offset.cpu()[:, ::4, ::4], as_t(gold["img_offset_s4"])) < 3e-5
    assert rel_err(S.sig.rss(warped).cpu(), as_t(gold["img_warped_rss"])) < 3e-5
    assert e32 < 1e-4 and e64 < max(1e-4, 2 * e_ref)
    assert abs(loss_sim.item() - float(gold["loss_sim"])) < 2e-5
    # sensitivity maps: per-coil checksums (sum re, sum im, L2 over the plane; |S| <= 1 so sums are O(1e5))
    sens = net_R.sens_net(k_samp, int(w * sp * 0.32)).cpu()
    got = np.stack([sens.real.double().sum((0, 2, 3)).numpy(), sens.imag.double().sum((0, 2, 3)).numpy(),
                    sens.abs().double().pow(2).sum((0, 2, 3)).sqrt().numpy()], 1)
    want = gold["sens_checksums"]
    assert np.all(np.abs(got[:, 2] - want[:, 2]) < 1e-4 * want[:, 2]), (got[:, 2], want[:, 2])
    assert np.all(np.abs(got[:, :2] - want[:, :2]) < 2e-3 * want[:, 2:3])
    cs = gold["cascade_checksums"]
    for j in range(12):
        xj = S.ops.owner_arena(net_R).get(f"cas{j}.xout", (n, c, h, w), torch.device(DEV), dtype=torch.complex64).cpu()
        # ortho transforms: the k-space L2 norm is the image-domain L2 norm
        assert abs(xj.abs().double().pow(2).sum().sqrt().item() - cs[j, 2]) < 1e-4 * cs[j, 2], j


def test_multicoil_two_cascade_train_step_golden(S):
    """[round 2] Config-4 shape, 2 cascades: a full 'Rec' step (train-mode BatchNorm on 30-channel input, sensitivity-map
    gradients through every coil, the W % 4 != 0 weight-gradient form at the 46-wide level) against the reference's
    losses and gradient digests."""
    gold = load_golden("multicoil_640x368.npz")
    n, c, h, w, sp = 1, 15, 640, 368, 0.125
    cfg = S.base.Config(sparsity=sp, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                        weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2)
    net = S.model.CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, sp, 0)
    _load(S, net.net_T, 3335)
    _load(S, net.net_R, 3336)
    net.to(DEV).train()
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=3334)
    net.set_input(g(img_full), g(img_aux))
    net.loss_all = 0
    net.forwardT()
    net.forwardR()
    assert rel_err(net.img_rec.cpu(), as_t(gold["train2.img_rec"])) < 1e-4
    for k_ in ("loss_sim", "loss_smooth", "loss_all"):
        want = float(gold["train2." + k_])
        got = (net.loss_all if k_ == "loss_all" else getattr(net, k_)).item()
        assert abs(got - want) < 1e-4 * max(1.0, abs(want)), (k_, got, want)
    for o in (net.optim_R, net.optim_T):
        o.zero_grad()
    with S.ops.wgrad_overlap():
        net.backward(train_T=True)
    torch.cuda.synchronize()
    for tag, mod in (("R", net.net_R), ("T", net.net_T)):
        wn, name, pe = _digest_errors_r2(S, [(nm, p.grad) for nm, p in mod.named_parameters()], gold, f"train2.grad.{tag}.")
        print(f"multi-coil net_{tag}: worst per-tensor norm error {wn:.2e} ({name}), probe-estimated relative L2 {pe:.2e}")
        # measured 1.5e-3 / 1.1e-3 (net_R), 2.6e-3 / 3.4e-3 (net_T: train-mode BatchNorm on one slice); bars = 3x measured
        # 
        bar_wn, bar_pe = (4.5e-3, 3.5e-3) if tag == "R" else (8e-3, 1.0e-2)
        assert wn < bar_wn and pe < bar_pe, (tag, wn, name, pe)


@pytest.mark.parametrize("mode,bar_db", [("bf16x2", 70.0), ("bf16", 30.0)])
def test_conv_precision_modes_e2e_psnr(S, mode, bar_db):
    """[round 2] The 12-cascade network at 320 x 320 with the convolutions in a narrow-precision mode, judged by PSNR against the
    fp32-equivalent output of the same network ('bf16 / fp8 configs cannot meet 1e-4; judge those by
    PSNR').  Also one optimisation step of the small model with cfg.use_amp (the reference's AMP seam, model.py:83-87)."""
    gold = load_golden("e2e_full_320.npz")
    w = 320
    img_full, img_aux = S.synth.phantom_pair(1, 1, w, w, seed=1234)
    pruned = as_t(gold["pruned"])
    net_R = S.varnet.VarNet(num_cascades=12, sens_chans=8, sens_pools=4, chans=18, pools=4, use_ref=True)
    _load(S, net_R, 1236)
    net_R.to(DEV).eval()
    keep = (~pruned).float().to(DEV)
    k_samp = S.ops.fft2c(g(img_full), colmask_out=keep)
    warped = g(as_t(gold["img_warped"]))
    try:
        with torch.no_grad():
            ref = net_R(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32)).cpu()
            with S.ops.conv_precision(mode):
                rec = net_R(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32)).cpu()
        assert rel_err(ref, as_t(gold["img_rec"])) < 1e-4
        psnr, rel = _psnr(ref, rec), rel_err(rec, ref)
        print(f"{mode}: PSNR vs the fp32-equivalent output {psnr:.1f} dB, rel-L2 {rel:.2e}")
        assert psnr > bar_db
        cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=80, coils=3, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                            weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=True, conv_dtype=mode, num_cascades=2,
                            chans=18, sens_chans=8, pools=2, sens_pools=2)
        net = S.model.CSModel(cfg)
        net.net_mask.pruned = S.synth.equispaced_pruned(80, 0.25, 0)
        _load(S, net.net_T, 41)
        _load(S, net.net_R, 42)
        net.to(DEV).train()
        f, a_ = S.synth.phantom_pair(2, 3, 48, 80, seed=40)
        before = [p.detach().clone() for p in net.net_R.parameters()]
        net.set_input(g(f), g(a_))
        net.update()
        assert S.ops.lib().query("san_get_conv_precision") == 3          # update() restores the process-wide mode
        after = list(net.net_R.parameters())
        assert all(torch.isfinite(p).all() for p in after) and any(not torch.equal(p, q) for p, q in zip(after, before))
    finally:
        S.ops.set_conv_precision("bf16x3")


def test_mixed_backward_precision_full_320(S):
    """[round 2] cfg.bwd_dtype = 'bf16x2' (fp32-equivalent forward, backward convolutions on two bf16 parts) on the damped full-size
    fixture: the forward is untouched (same bars as test_train_step_full_320_golden) and the gradients stay within the
    same 3x-the-reference's-own-noise bars against the float64 arbiter."""
    gold = load_golden("train_full_320.npz")
    tag, n, c, h, w = "damped", 2, 1, 320, 320
    cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                        weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=12, bwd_dtype="bf16x2")
    net = S.model.CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    _load(S, net.net_T, 2235)
    net.net_R.load_state_dict(S.synth.fill_params(_shapes(net.net_R), seed=2236, damp=0.1))
    net.to(DEV).train()
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=2234)
    net.set_input(g(img_full), g(img_aux))
    net.loss_all = 0
    net.forwardT()
    net.forwardR()
    assert rel_err(net.img_rec.cpu(), as_t(gold[f"{tag}.f32.img_rec"])) < 1e-4
    for o in (net.optim_R, net.optim_T):
        o.zero_grad()
    try:
        with S.ops.wgrad_overlap(), S.ops.conv_precision(net.bwd_dtype):
            net.backward(train_T=True)
        torch.cuda.synchronize()
    finally:
        S.ops.set_conv_precision("bf16x3")
    for nt, mod in (("R", net.net_R), ("T", net.net_T)):
        named = [(nm, p.grad) for nm, p in mod.named_parameters()]
        floor = float(gold[f"{tag}.ref32_vs_ref64.grad.{nt}"])
        wn64, name64, pe64 = _digest_errors_r2(S, named, gold, f"{tag}.f64.grad.{nt}.")
        print(f"mixed backward, net_{nt}: probe-estimated relative L2 vs ref64 {pe64:.2e}, worst per-tensor norm {wn64:.2e} "
              f"({name64}); reference fp32-vs-fp64 {floor:.2e}")
        assert pe64 < max(3.0 * floor, 5e-4) and wn64 < max(3.0 * floor, 5e-4)


def test_fp8_mode_e2e_psnr_and_train_step(S):
    """[round 2] Config 5: the 12-cascade network at 320 x 320 with fp8 e4m3 forward convolutions (fp32 FFT / DC / norms / losses),
    judged by PSNR against the fp32-equivalent output of the same network, and one 'Rec' optimisation step with
    cfg.conv_dtype = 'fp8' (forward fp8, data / weight gradients bf16)."""
    gold = load_golden("e2e_full_320.npz")
    w = 320
    img_full, img_aux = S.synth.phantom_pair(1, 1, w, w, seed=1234)
    pruned = as_t(gold["pruned"])
    net_R = S.varnet.VarNet(num_cascades=12, sens_chans=8, sens_pools=4, chans=18, pools=4, use_ref=True)
    _load(S, net_R, 1236)
    net_R.to(DEV).eval()
    keep = (~pruned).float().to(DEV)
    k_samp = S.ops.fft2c(g(img_full), colmask_out=keep)
    warped = g(as_t(gold["img_warped"]))
    try:
        with torch.no_grad():
            ref = net_R(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32)).cpu()
            with S.ops.conv_precision("fp8"):
                rec = net_R(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32)).cpu()
            with S.ops.conv_precision("bf16"):
                rec16 = net_R(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32)).cpu()
        psnr, rel = _psnr(ref, rec), rel_err(rec, ref)
        print(f"fp8: PSNR vs the fp32-equivalent output {psnr:.1f} dB, rel-L2 {rel:.2e} (bf16: {_psnr(ref, rec16):.1f} dB)")
        assert torch.isfinite(rec).all()
        assert psnr > 20.0          # measured 30.2 dB (bf16: 45.8 dB); random-init weights amplify rounding noise through 12 cascades
        f, a_ = S.synth.phantom_pair(2, 3, 48, 80, seed=40)

        def one_step(conv_dtype):
            cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=80, coils=3, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                                weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, conv_dtype=conv_dtype,
                                num_cascades=2, chans=18, sens_chans=8, pools=2, sens_pools=2)
            net = S.model.CSModel(cfg)
            net.net_mask.pruned = S.synth.equispaced_pruned(80, 0.25, 0)
            _load(S, net.net_T, 41)
            _load(S, net.net_R, 42)
            net.to(DEV).train()
            before = [p.detach().clone() for p in net.net_R.parameters()]
            net.set_input(g(f), g(a_))
            net.update()
            assert S.ops.lib().query("san_get_conv_precision") == 3          # update() restores the process-wide mode
            after = list(net.net_R.parameters())
            assert all(torch.isfinite(p).all() for p in after) and any(not torch.equal(p, q) for p, q in zip(after, before))
            return torch.cat([p.grad.flatten() for p in net.net_R.parameters() if p.grad is not None]).double().cpu()

        g8, g32 = one_step("fp8"), one_step("bf16x3")
        cos = float((g8 * g32).sum() / (g8.norm() * g32.norm()))
        print(f"fp8 train step: gradient cosine vs the fp32-equivalent step {cos:.4f}")
        assert cos > 0.8            # the fp8 step descends along the fp32 gradient: measured 0.939
    finally:
        S.ops.set_conv_precision("bf16x3")


# ------------------------------------------------------------------------------------------- the reference's smoke idiom
@pytest.mark.parametrize("tag,shape", [("32", (2, 1, 32, 32)), ("48x80c3", (2, 3, 48, 80))])
def test_reference_smoke_idiom_matches_direct_chain(S, tag, shape):
    """[round 3] ``result = varnet(...); ssimloss(result, target).backward()`` (the reference's own smoke block, varnet.py:546-560):
    every p.grad (a) equals the direct VarNet.backward chain bit for bit and (b) matches the gradients the reference
    produced for the same weights and inputs."""
    gold = load_golden(f"e2e_small_{tag}.npz")
    n, c, h, w = shape
    pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    k_samp, warped = g(as_t(gold["train.img_k_sampled"], True)), g(as_t(gold["train.img_warped"]))
    full_rss = g(as_t(gold["train.img_full_rss"]))

    def build():
        net = S.varnet.VarNet(num_cascades=2, sens_chans=2, sens_pools=2, chans=4, pools=2, use_ref=True)
        _fill(S, net, 42)
        return net.to(DEV).train()

    net_a = build()
    result = net_a(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32))
    assert result.grad_fn is not None and result.requires_grad
    S.ssim.ssimloss(full_rss, result).backward()
    net_b = build()
    rec = net_b(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32))
    net_b.backward(S.ops.ssim_loss_bwd(full_rss, rec.detach(), 1.0), want_ref_grad=False)
    assert torch.equal(result, rec)
    worst = 0.0
    for (name, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
        assert pa.grad is not None and torch.equal(pa.grad, pb.grad), name
        want = as_t(gold["grad.R." + name])
        scale = want.abs().max().item()
        if scale > 1e-12:
            worst = max(worst, (pa.grad.cpu() - want).abs().max().item() / scale)
    print("autograd route: worst relative gradient error vs the reference", worst)
    assert worst < 2e-4, worst
    # a second backward through the same (stale once a new forward ran) graph is refused
    result2 = net_a(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32))
    net_a(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32))
    with pytest.raises(RuntimeError, match="no longer the module's latest"):
        S.ssim.ssimloss(full_rss, result2).backward()


@pytest.mark.parametrize("shape,weight_sim", [((2, 1, 32, 32), 1.0), ((2, 3, 48, 80), 0.37)])
def test_loss_all_backward_matches_update_chain_bitwise(S, shape, weight_sim):
    """[round 3] CSModel 'Rec': forwardT(); forwardR(); loss_all.backward() (the reference's model.py:203-214 idiom, through
    autograd: SSIM -> VarNet -> ref -> warp -> grid -> offset (+ smoothness) -> alignment U-Net) fills every p.grad of
    BOTH networks with the same bits as the direct CSModel.backward chain, also with loss weights != 1."""
    n, c, h, w = shape
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    res = []
    for route in ("autograd", "direct"):
        net = _rec_model(S, w, c, weight_sim=weight_sim).to(DEV).train()
        net.set_input(g(img_full), g(img_aux))
        net.loss_all = 0
        net.forwardT()
        net.forwardR()
        for o in (net.optim_R, net.optim_T):
            o.zero_grad()
        if route == "autograd":
            assert net.loss_all.grad_fn is not None
            net.loss_all.backward()
        else:
            net.backward(train_T=True)
        torch.cuda.synchronize()
        res.append((_grads(net), net.loss_all.detach().clone(), [b.detach().clone() for b in net.net_T.buffers()]))
    (ga, la, ba), (gd, ld, bd) = res
    assert torch.equal(la, ld)
    assert all(torch.equal(x, y) for x, y in zip(ba, bd))
    bad = [i for i, (x, y) in enumerate(zip(ga, gd)) if not torch.equal(x, y)]
    assert not bad, f"{len(bad)} of {len(ga)} parameter gradients differ between loss_all.backward() and the direct chain"
    assert any(x.abs().max().item() > 0 for x in ga)


def test_autograd_route_trains_like_update(S):
    """[round 3] Three optimisation steps written the reference's way (zero_grad; loss_all.backward(); optim.step()) give
    bit-identical parameters to three CSModel.update() calls."""
    n, c, h, w = 2, 1, 32, 32
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    outs = []
    for route in ("autograd", "update"):
        net = _rec_model(S, w, c).to(DEV).train()
        for _ in range(3):
            net.set_input(g(img_full), g(img_aux))
            if route == "update":
                net.update()
                continue
            net.loss_all = 0
            net.forwardT()
            net.forwardR()
            net.optim_T.zero_grad()
            net.optim_R.zero_grad()
            net.loss_all.backward()
            net.optim_T.step()
            net.optim_R.step()
            del net.loss_all
        torch.cuda.synchronize()
        outs.append([p.detach().clone() for m in (net.net_R, net.net_T) for p in m.parameters()])
    assert all(torch.equal(x, y) for x, y in zip(*outs))


def test_regime_none_through_autograd(S):
    """[round 3] Regime 'None' (model.py:195-204): forwardT under no_grad, only net_R trains; loss_all.backward() leaves net_T's
    gradients untouched and equals the direct chain."""
    n, c, h, w = 2, 1, 32, 32
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    res = []
    for route in ("autograd", "direct"):
        net = _rec_model(S, w, c).to(DEV).train()
        net.set_input(g(img_full), g(img_aux))
        net.loss_all = 0
        with torch.no_grad():
            net.forwardT()
        net.loss_all = 0
        net.forwardR()
        net.optim_R.zero_grad()
        net.optim_T.zero_grad()
        if route == "autograd":
            assert not net.img_warped.requires_grad
            net.loss_all.backward()
        else:
            net.backward(train_T=False)
        torch.cuda.synchronize()
        res.append(_grads(net))
        assert all(p.grad.abs().max().item() == 0 for p in net.net_T.parameters())
    assert all(torch.equal(x, y) for x, y in zip(*res))


def test_vis_images_survive_the_next_step(S):
    """[round 3] get_vis('images') hands out tensors the next step does not overwrite (the reference returns fresh tensors)."""
    n, c, h, w = 2, 1, 32, 32
    net = _rec_model(S, w, c).to(DEV).train()
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    net.set_input(g(img_full), g(img_aux))
    net.update()
    vis = net.get_vis("images")["images"]
    keep = {k: v.clone() for k, v in vis.items()}
    img2, aux2 = S.synth.phantom_pair(n, c, h, w, seed=78)
    net.set_input(g(img2), g(aux2))
    net.update()
    torch.cuda.synchronize()
    assert {"img_rec", "img_warped", "img_full_rss"} <= set(vis)
    for k, v in vis.items():
        assert torch.equal(v, keep[k]), k


def test_alignment_backward_in_eval_mode_vs_oracle_autograd(S):
    """[round 3] Backward through the alignment network in eval mode (BatchNorm on running statistics): every
    parameter gradient incl. gamma / beta against oracle autograd."""
    n, c, h, w = 2, 1, 32, 48
    st = S.cross.SpatialTransformer(c)
    p = S.synth.fill_params([(k, tuple(v.shape)) for k, v in st.state_dict().items()], seed=61)
    st.load_state_dict(p)
    st.to(DEV).eval()
    moving, fixed = philox("ev.m", (n, c, h, w), lo=0.0, hi=1.0), philox("ev.f", (n, c, h, w), lo=0.0, hi=1.0)
    wgt = philox("ev.w", (n, h, w, 2))
    off, grid = st(g(moving), g(fixed))
    assert off.grad_fn is not None
    (off * g(wgt)).sum().backward()
    p64 = {k: v.double().requires_grad_(v.dtype.is_floating_point) for k, v in p.items()}
    off64, _ = S.O.spatial_transformer_forward(p64, moving.double(), fixed.double(), training=False)
    (off64 * wgt.double()).sum().backward()
    assert rel_err(off.detach().cpu(), off64.detach().float()) < 1e-4
    worst, wname = 0.0, ""
    for name, prm in st.named_parameters():
        want = p64[name].grad
        scale = want.abs().max().item()
        if scale < 1e-9:
            continue
        err = (prm.grad.cpu().double() - want).abs().max().item() / scale
        if err > worst:
            worst, wname = err, name
    print("eval-mode alignment backward: worst relative gradient error", worst, wname)
    assert worst < 2e-3, (worst, wname)


def test_train_step_bench_batch_n8_golden(S):
    """[round 3] VERDICT r2 #7: the batch the bench is quoted on -- one 'Rec' training step at N = 8, 320 x 320, 12 cascades, chans 18
    (BASELINE configs[1]; model.py:206-216), default kernel mix incl. the side stream, 'damped' weights (cascade output
    convolutions x 0.1: a trained-like network) -- against the reference's fp32 step with its own fp64 step as arbiter
    (tests/golden/train_n8_320.npz, make_golden.py `train_n8`).  Image: every second row / column + per-slice norms;
    gradients: per-tensor L2 norms + 16 probes per tensor, both networks."""
    gold = load_golden("train_n8_320.npz")
    assert S.ops.USE_BF16X3[0] and S.ops.WGRAD_OVERLAP[0]
    n, c, h, w = 8, 1, 320, 320
    cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                        weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=12)
    net = S.model.CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    _fill(S, net.net_T, 4235)
    _fill(S, net.net_R, 4236, damp=0.1)
    net.to(DEV).train()
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=4234)
    net.set_input(g(img_full), g(img_aux))
    net.loss_all = 0
    net.forwardT()
    net.forwardR()
    rec = net.img_rec.detach().cpu()
    e_ref = float(gold["ref32_vs_ref64.img_rec"])
    e32 = rel_err(rec[:, :, ::2, ::2], as_t(gold["f32.img_rec_s2"]))
    e64 = rel_err(rec[:, :, ::2, ::2].double(), as_t(gold["f64.img_rec_s2"]))
    print(f"[n8] train-mode rec (every 2nd row / column): hip-vs-ref32 {e32:.2e}, hip-vs-ref64 {e64:.2e}, ref32-vs-ref64 {e_ref:.2e}")
    assert e64 < max(1e-4, 2 * e_ref)                   # north_star's 1e-4, or twice the reference's own fp32 distance
    assert e32 < max(1e-4, 3 * e_ref)
    l2 = rec.double().pow(2).sum((1, 2, 3)).sqrt().numpy()
    assert np.all(np.abs(l2 - gold["f64.img_rec_l2"]) < 1e-4 * gold["f64.img_rec_l2"]), (l2, gold["f64.img_rec_l2"])
    wl2 = net.img_warped.detach().cpu().double().pow(2).sum((1, 2, 3)).sqrt().numpy()
    assert np.all(np.abs(wl2 - gold["f32.img_warped_l2"]) < 3e-5 * gold["f32.img_warped_l2"])
    for k_ in ("loss_sim", "loss_smooth", "loss_all"):
        want = float(gold[f"f32.{k_}"])
        got = (net.loss_all if k_ == "loss_all" else getattr(net, k_)).item()
        assert abs(got - want) < max(2e-5, 2 * e_ref) * max(1.0, abs(want)), (k_, got, want)
    for o in (net.optim_R, net.optim_T):
        o.zero_grad()
    net.backward(train_T=True)
    torch.cuda.synchronize()
    for nt, mod in (("R", net.net_R), ("T", net.net_T)):
        named = [(nm, p.grad) for nm, p in mod.named_parameters()]
        floor = float(gold[f"ref32_vs_ref64.grad.{nt}"])
        d32, d64 = gold[f"f32.grad.{nt}.probes"], gold[f"f64.grad.{nt}.probes"]
        ne = gold[f"f64.grad.{nt}.numel"][:, None]
        floor_probe = float((((d32 - d64) ** 2 * ne / 16.0).sum() / (gold[f"f64.grad.{nt}.l2"] ** 2).sum()) ** 0.5)
        l2a, l2b = gold[f"f32.grad.{nt}.l2"], gold[f"f64.grad.{nt}.l2"]
        big = l2b > 1e-3 * l2b.max()
        floor_norm = float((np.abs(l2a - l2b) / np.maximum(l2b, 1e-30))[big].max())
        wn64, name64, pe64 = _digest_errors_r3(S, named, gold, f"f64.grad.{nt}.")
        wn32, name32, pe32 = _digest_errors_r3(S, named, gold, f"f32.grad.{nt}.")
        print(f"[n8] net_{nt}: per-tensor norm error vs ref64 {wn64:.2e} ({name64}), vs ref32 {wn32:.2e}; probe-estimated relative "
              f"L2 vs ref64 {pe64:.2e}, vs ref32 {pe32:.2e}; reference fp32-vs-fp64: exact {floor:.2e}, probe-estimated "
              f"{floor_probe:.2e}, worst per-tensor norm {floor_norm:.2e}")
        assert pe64 < max(3.0 * max(floor, floor_probe), 5e-4), (nt, pe64, floor, floor_probe)
        assert wn64 < max(3.0 * max(floor, floor_norm), 5e-4), (nt, wn64, name64, floor, floor_norm)
    for k_ in gold.files:
        if k_.startswith("f32.bn_after.T."):
            got = dict(net.net_T.named_buffers())[k_[len("f32.bn_after.T."):]]
            assert torch.allclose(got.cpu(), as_t(gold[k_]), rtol=2e-4, atol=2e-6), k_


def test_narrow_precision_psnr_on_trained_like_weights(S):
    """[round 3] VERDICT r2 #7: bf16 / fp8 convolutions judged where the judgement means something -- on the 'damped' weight set
    (cascade output convolutions x 0.1, i.e. every cascade a small correction as in a trained network; random-init
    weights amplify rounding noise through the 12 cascades and read 46 / 30 dB) at the bench batch: PSNR of the N = 8
    reconstruction against the fp32-equivalent output of the same weights AND against the reference's float64 output.
    Measured: bf16 44.9 dB (43.5 vs fp64), fp8 30.2 dB (29.4 vs fp64); the fp32-equivalent mode itself reads 98.6 dB vs fp64.
    That is what one bf16 / e4m3 operand rounding per convolution costs through 13 U-Nets + the alignment net here -- the
    damping does not change it (random-init weights: 45.8 / 30.2 dB), so the review's expectation of >= 55 / >= 40 dB does
    not hold for these formats; the bars sit 5 dB below the measured values."""
    gold = load_golden("train_n8_320.npz")
    n, c, h, w = 8, 1, 320, 320
    cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                        weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=12)
    net = S.model.CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    _fill(S, net.net_T, 4235)
    _fill(S, net.net_R, 4236, damp=0.1)
    net.to(DEV).train()                                    # train-mode BatchNorm, as in the fixture
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=4234)
    ref64 = as_t(gold["f64.img_rec_s2"])
    out = {}
    try:
        for mode in ("bf16x3", "bf16", "fp8"):
            net.conv_dtype = mode
            with torch.no_grad(), S.ops.conv_precision(mode):
                net.set_input(g(img_full), g(img_aux))
                net.loss_all = 0
                net.forwardT()
                net.forwardR()
            out[mode] = net.img_rec.detach().cpu().clone()
    finally:
        S.ops.set_conv_precision("bf16x3")
    base = out["bf16x3"]
    res = {m_: (_psnr(base, out[m_]), _psnr(ref64, out[m_][:, :, ::2, ::2].double())) for m_ in ("bf16", "fp8")}
    print("narrow precision on damped weights, PSNR vs fp32-equivalent / vs the reference's fp64:", res,
          "| fp32-equivalent vs fp64:", _psnr(ref64, base[:, :, ::2, ::2].double()))
    BARS = {"bf16": 40.0, "fp8": 25.0}                      # measured 44.9 / 30.2 dB (vs fp64: 43.5 / 29.4)
    for m_, (p32, p64) in res.items():
        assert p32 > BARS[m_] and p64 > BARS[m_] - 1.0, (m_, p32, p64)


# ------------------------------------------------------------------------------------------- bench batch, eval mode
def test_eval_bench_batch_n8_golden(S):
    """[round 4] VERDICT r3 #10: the bench batch in EVAL mode -- N = 8 slices of 320 x 320, 12 cascades, chans 18 -- against the
    reference's fp32 forward with its fp64 run as arbiter (tests/golden/eval_n8_320.npz, made by make_golden.py eval_n8).  Per
    slice: within max(1e-4, 2 x the reference's own fp32-fp64 distance of that slice) of BOTH references (the reference itself
    is 1.9e-4 from its fp64 on slice 2, 2.9-5.2e-5 elsewhere); whole slices 0 and 5, 64 probed pixels, sum and L2 of every slice."""
    from conftest import load_golden, as_t
    from spatialalignmentnetwork_amd import cross, varnet, signal_utils, ssimloss
    gold = load_golden("eval_n8_320.npz")
    n, c, h, w = 8, 1, 320, 320
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=1234)
    pruned = as_t(gold["pruned"])
    net_T = cross.SpatialTransformer(1)
    net_R = varnet.VarNet(num_cascades=12, sens_chans=8, sens_pools=4, chans=18, pools=4, use_ref=True)
    _fill(S, net_T, 1235)
    _fill(S, net_R, 1236)
    net_T.to(DEV).eval()
    net_R.to(DEV).eval()
    with torch.no_grad():
        keep = (~pruned).float().to(DEV)
        k_samp = S.ops.fft2c(g(img_full), colmask_out=keep)
        samp = signal_utils.ifft2(k_samp)
        aux_abs = S.ops.cabs(g(img_aux))
        offset, grid = net_T(aux_abs, S.ops.cabs(samp))
        warped = net_T.warp(aux_abs, grid)
        rec = net_R(k_samp, (~pruned).to(DEV), warped, int(w * 0.25 * 0.32))
    torch.cuda.synchronize()
    rec = rec.cpu().double().reshape(n, -1)
    idx = torch.from_numpy(gold["probe_idx"]).long()
    floor = torch.from_numpy(gold["ref_f32_vs_f64_rel_per_slice"])
    worst = 0.0
    for tag in ("f32", "f64"):
        l2 = torch.from_numpy(gold[f"rec_{tag}.l2"])
        # whole slices
        for sl in (0, 5):
            want = torch.from_numpy(gold[f"rec_{tag}.slice{sl}"]).double().reshape(-1)
            e = ((rec[sl] - want).norm() / want.norm()).item()
            worst = max(worst, e)
            assert e < max(1e-4, 2 * floor[sl].item()), (tag, sl, e)
        # every slice: probes (relative to the slice's RMS), sum and L2
        rms = l2 / (h * w) ** 0.5
        probe = torch.from_numpy(gold[f"rec_{tag}.probe"])
        bar = torch.clamp(2 * floor, min=1e-4)
        perr = ((rec[:, idx] - probe).abs().max(1).values / rms)
        assert torch.all(perr < 40 * bar), (tag, perr)            # a single pixel against the slice RMS: measured <= 6e-4
        assert torch.all(((rec.norm(dim=1) - l2).abs() / l2) < bar), tag
        assert torch.all(((rec.sum(1) - torch.from_numpy(gold[f"rec_{tag}.sum"])).abs() / (l2 * (h * w) ** 0.5)) < bar), tag
    wl2 = torch.from_numpy(gold["warped_f32.l2"])
    assert torch.all(((warped.cpu().double().reshape(n, -1).norm(dim=1) - wl2).abs() / wl2) < 3e-5)
    print(f"eval N = 8: worst whole-slice rel-L2 vs the references {worst:.2e}")


# ------------------------------------------------------------------------------------------- gradients on the SHIPPED kernels
@pytest.mark.parametrize("tag,shape", [("32", (2, 1, 32, 32)), ("48x80c3", (2, 3, 48, 80))])
def test_full_rec_step_gradients_elementwise_on_shipped_kernels(S, tag, shape):
    """[round 4] VERDICT r3 #10: EVERY parameter gradient of a 'Rec' step, element-wise, against the reference's own gradients
    (tests/golden/e2e_small_*.npz) with the SHIPPED kernel mix (fp16-part matrix-core convolutions, data and weight gradients,
    weight gradients on the side stream) -- not the fp32 kernels the round-1 test switches to.  Bar per tensor: max-abs error
    <= 3e-4 of the tensor's largest reference gradient; measured 1.4e-5 / 5.7e-5 (32 x 32: net_R / net_T) and 2.4e-5 / 5.2e-5
    (48 x 80, 3 coils) -- no tensor needs to be skipped: the kink-dominated BatchNorm tensors that made round 1 switch kernels
    (8e-4 on the fp32 path) land at 1e-5 here."""
    from conftest import load_golden, as_t
    gold = load_golden(f"e2e_small_{tag}.npz")
    n, c, h, w = shape
    cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                        weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=4,
                        sens_chans=2, pools=2, sens_pools=2)
    net = S.model.CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    _fill(S, net.net_T, 41)
    _fill(S, net.net_R, 42)
    net.to(DEV).train()
    assert S.ops.USE_BF16X3[0] and S.ops.current_precision() == "bf16x3"
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    net.set_input(g(img_full), g(img_aux))
    net.loss_all = 0
    net.forwardT()
    net.forwardR()
    for o in (net.optim_R, net.optim_T):
        o.zero_grad()
    net.backward(train_T=True)
    torch.cuda.synchronize()
    for pre, mod in (("grad.R.", net.net_R), ("grad.T.", net.net_T)):
        worst, worst_name = 0.0, ""
        for name, prm in mod.named_parameters():
            want = as_t(gold[pre + name])
            scale = want.abs().max().item()
            got = prm.grad.cpu() if prm.grad is not None else torch.zeros_like(want)
            if scale < 1e-7:
                assert got.abs().max().item() < 1e-5, name
                continue
            err = (got - want).abs().max().item() / scale
            if err > worst:
                worst, worst_name = err, name
        print(f"{tag} {pre} shipped kernels: worst element-wise relative gradient error {worst:.2e} ({worst_name})")
        assert worst < 3e-4, (pre, worst, worst_name)
