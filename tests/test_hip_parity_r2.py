"""Round-2 GPU parity tests (through the C ABI): shapes that need padding, the single-layer fixtures, the CSModel
protocol, the config-2 train step at full size with every bf16x3 kernel on, config 4 (15 coils, 640 x 368) and the
data-parallel step.  Tolerances are written next to each assertion together with what was measured."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import as_t, cplx, philox, rel_err, load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def S():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spatialalignmentnetwork_amd import ops, synth, varnet, cross, unet, signal_utils, ssimloss, masks, model, basemodel
    from oracle import cpu_ref as O

    class NS:
        pass

    ns = NS()
    ns.ops, ns.synth, ns.varnet, ns.cross, ns.unet, ns.sig, ns.ssim = ops, synth, varnet, cross, unet, signal_utils, ssimloss
    ns.masks, ns.model, ns.base, ns.O = masks, model, basemodel, O
    return ns


def g(t):
    return t.to(DEV).contiguous()


def _shapes(m):
    return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]


def _load(S, m, seed):
    p = S.synth.fill_params(_shapes(m), seed=seed)
    m.load_state_dict(p)
    return p


def probe_idx(S, name, numel, k=16):
    return S.synth._rng("probe." + name, 0).integers(0, numel, k)


# ------------------------------------------------------------------ window copy / padding
def test_window_copy_modes(S):
    """san_window_copy_fwd against F.pad: zero pad, crop, reflect (bottom / right) and the reflect adjoint.  Exact."""
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


def test_normunet_pad_golden(S):
    """NormUnet at 50 x 70 (zero pad of the normalised image to 64 x 80, crop, un-normalise; varnet.py:275-332) forward
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
    """Bare U-Net at 25 x 35 with 2 pooling levels: the avg-pool drops odd rows / columns and the up path reflect-pads
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
    """VarNet (2 cascades, 2 coils, sensitivity net) on 50 x 70 slices, eval."""
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


def test_normunet_backward_with_constant_plane(S):
    """An all-zero slice in the batch has std == 0 on both planes: the reference stays finite (forward divides by
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


# ------------------------------------------------------------------ image-domain cascade boundary
@pytest.mark.parametrize("n,c,h,w", [(2, 1, 320, 320), (1, 3, 320, 320), (2, 3, 48, 80), (1, 2, 46, 368), (2, 1, 30, 45), (1, 1, 6, 320),
                                     (1, 2, 7, 320), (2, 1, 9, 368), (1, 15, 8, 368)])
def test_dc_rows_vs_kspace_formula(S, n, c, h, w):
    """san_dc_rows (one row-local launch per cascade on x = ifft2(k)) against the reference's k-space update
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


# ------------------------------------------------------------------ single layers
def test_varnetblock_step_and_sens_expand_golden(S, ops_golden):
    """One cascade with a real regulariser through VarNetBlock.forward, and the stand-alone sens_expand
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


@pytest.mark.parametrize("tag,cin,cout,shp", [("conv2d", 6, 16, (2, 6, 24, 40)), ("up", 16, 24, (2, 16, 12, 20)),
                                              ("down", 24, 16, (2, 24, 24, 40))])
def test_alignment_layers_golden(S, tag, cin, cout, shp):
    """The alignment backbone's Conv2d / Up / Down factories (unet.py:119-140) in eval and train mode, incl. the
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
    """set_input -> test() -> get_vis() -> save() -> load() on the GPU (model.py:89-121,265-321; basemodel.py:159-182)
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


# ------------------------------------------------------------------ full-size train step (config 2 shape)
def _digest_errors(S, named_grads, gold, pre):
    """Per-network relative L2 (from per-tensor norms and probes) of our gradients against a digest fixture."""
    names = [str(s) for s in gold[pre + "names"]]
    l2 = gold[pre + "l2"]
    grads = dict(named_grads)
    worst_norm, worst_name, num, den = 0.0, "", 0.0, 0.0
    for i, nm in enumerate(names):
        got = grads[nm].detach().double().reshape(-1).cpu()
        assert got.numel() == int(gold[pre + "numel"][i]), nm
        e = abs(got.norm().item() - float(l2[i])) / max(float(l2[i]), 1e-30)
        if float(l2[i]) > 1e-3 * float(l2.max()) and e > worst_norm:
            worst_norm, worst_name = e, nm
        pr = got[torch.from_numpy(probe_idx(S, nm, got.numel()))]
        want = torch.from_numpy(gold[pre + "probes"][i])
        num += ((pr - want) ** 2).sum().item() * got.numel() / 16.0         # probes as a 16-sample estimate of the tensor
        den += float(l2[i]) ** 2
    return worst_norm, worst_name, (num / den) ** 0.5


@pytest.mark.parametrize("tag,damp", [("raw", 1.0), ("damped", 0.1)])
def test_train_step_full_320_golden(S, tag, damp):
    """One 'Rec' training step at the bench shape (N = 2, 320 x 320, 12 cascades, chans 18; model.py:206-216) with the
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
        wn32, name32, pe32 = _digest_errors(S, named, gold, f"{tag}.f32.grad.{nt}.")
        wn64, name64, pe64 = _digest_errors(S, named, gold, f"{tag}.f64.grad.{nt}.")
        print(f"[{tag}] net_{nt}: per-tensor norm error vs ref32 {wn32:.2e} ({name32}), vs ref64 {wn64:.2e} ({name64}); "
              f"probe-estimated relative L2 vs ref32 {pe32:.2e}, vs ref64 {pe64:.2e}; reference fp32-vs-fp64: exact "
              f"{floor:.2e}, probe-estimated {floor_probe:.2e}, worst per-tensor norm {floor_norm:.2e}")
        # no further from the fp64 truth than 3x the reference's own fp32 run, measured with the same estimators
        assert pe64 < max(3.0 * max(floor, floor_probe), 2e-3), (nt, pe64, floor, floor_probe)
        assert wn64 < max(3.0 * max(floor, floor_norm), 2e-3), (nt, wn64, name64, floor, floor_norm)
    if tag == "raw":
        for k_ in gold.files:
            if k_.startswith("f32.bn_after.T."):
                got = dict(net.net_T.named_buffers())[k_[len("f32.bn_after.T."):]]
                assert torch.allclose(got.cpu(), as_t(gold[k_]), rtol=2e-4, atol=2e-6), k_


def test_cascade_checksums_full_320(S):
    """Per-cascade k-space checksums (sum re, sum im, L2) of the 12-cascade network at 320 x 320 against the
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


# ------------------------------------------------------------------ config 4: multi-coil 640 x 368 x 15
def _multicoil_nets(S, num_cascades, seed):
    net_T = S.cross.SpatialTransformer(15)
    net_R = S.varnet.VarNet(num_cascades=num_cascades, sens_chans=8, sens_pools=4, chans=18, pools=4, use_ref=True)
    _load(S, net_T, seed + 1)
    _load(S, net_R, seed + 2)
    return net_T.to(DEV), net_R.to(DEV)


def test_e2e_multicoil_640x368_golden(S):
    """BASELINE config 4: one 640 x 368 slice, 15 coils, 8x equispaced mask (46 kept columns, 14 low frequencies),
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
    """Config-4 shape, 2 cascades: a full 'Rec' step (train-mode BatchNorm on 30-channel input, sensitivity-map
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
        wn, name, pe = _digest_errors(S, [(nm, p.grad) for nm, p in mod.named_parameters()], gold, f"train2.grad.{tag}.")
        print(f"multi-coil net_{tag}: worst per-tensor norm error {wn:.2e} ({name}), probe-estimated relative L2 {pe:.2e}")
        # measured 1.5e-3 / 1.1e-3 (net_R), 2.6e-3 / 3.4e-3 (net_T: train-mode BatchNorm on one slice); bars = 3x measured
        # 
        bar_wn, bar_pe = (4.5e-3, 3.5e-3) if tag == "R" else (8e-3, 1.0e-2)
        assert wn < bar_wn and pe < bar_pe, (tag, wn, name, pe)


# ------------------------------------------------------------------ data parallel (two ranks on one GPU)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_cfg(S, w):
    return S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=1, reg="None", mask="equispaced", weight_smooth=1000.0,
                         weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=4,
                         sens_chans=2, pools=2, sens_pools=2)


def _dp_worker(rank, world, port, path):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import types
    from spatialalignmentnetwork_amd import basemodel, dist as sdist, synth
    from spatialalignmentnetwork_amd.model import CSModel
    d = sdist.init("gloo")                                # gradients staged through the host: both ranks share cuda:0
    h, w = 48, 80
    torch.manual_seed(100 + rank)                         # replicas start DIFFERENT: update() must sync them from rank 0
    net = CSModel(_dp_cfg(types.SimpleNamespace(base=basemodel), w))
    net.net_mask.pruned = synth.equispaced_pruned(w, 0.25, 0)
    if rank == 0:
        for sub, sd in (("net_T", 41), ("net_R", 42)):
            m = getattr(net, sub)
            m.load_state_dict(synth.fill_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=sd))
    net.to("cuda:0").train()
    net.net_T.eval()                                      # frozen alignment net on running statistics: shards == full batch
    img_full, img_aux = synth.phantom_pair(2, 1, h, w, seed=40)
    lo, hi = sdist.shard_bounds(2, rank, world)
    out = {}
    for step in range(2):
        net.set_input(img_full[lo:hi].to("cuda:0").contiguous(), img_aux[lo:hi].to("cuda:0").contiguous())
        net.update()
        if step == 0:
            out["grad_sum"] = net.optim_R.bucket().flat.cpu().clone()      # after the all-reduce (sum over ranks)
    torch.cuda.synchronize()
    out["params"] = {k: v.cpu() for k, v in net.net_R.state_dict().items()}
    out["T"] = {k: v.cpu() for k, v in net.net_T.state_dict().items()}
    torch.save(out, f"{path}/rank{rank}.pt")
    d.barrier()
    d.destroy_process_group()


def test_update_data_parallel_two_ranks(S, tmp_path):
    """CSModel.update()'s data-parallel branch (flat-buffer all-reduce, 1/world inside AdamW, replica sync from rank 0)
    with two processes sharing this GPU over gloo: both ranks end with bit-identical parameters although they were
    constructed from different RNG streams; the all-reduced gradient equals 2x the whole-batch gradient of a
    single-process run (mean loss over 2 slices = mean of the two shard losses) to rounding."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert all(torch.equal(a["params"][k], b["params"][k]) for k in a["params"]), "replicas diverged"
    assert all(torch.equal(a["T"][k], b["T"][k]) for k in a["T"]), "rank 1 did not receive rank 0's alignment net"
    assert torch.equal(a["grad_sum"], b["grad_sum"])
    h, w = 48, 80
    net = S.model.CSModel(_dp_cfg(S, w))
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    _load(S, net.net_T, 41)
    _load(S, net.net_R, 42)
    net.to(DEV).train()
    net.net_T.eval()
    img_full, img_aux = S.synth.phantom_pair(2, 1, h, w, seed=40)
    net.set_input(g(img_full), g(img_aux))
    net.update()
    whole = net.optim_R.bucket().flat.cpu()
    err = rel_err(a["grad_sum"] / 2.0, whole)
    print("data-parallel averaged gradient vs whole-batch gradient, relative L2:", err)
    assert err < 1e-4
    net.set_input(g(img_full), g(img_aux))
    net.update()
    # parameters after two AdamW steps: each step moves a weight by ~lr (sign-like for the first steps), so compare the
    # DISPLACEMENT from the initial weights norm-wise (elements whose gradient is ~0 may step in opposite directions)
    init = S.synth.fill_params(_shapes(net.net_R), seed=42)
    num = den = 0.0
    for k, v in net.net_R.state_dict().items():
        d_ref, d_dp = v.cpu().double() - init[k].double(), a["params"][k].double() - init[k].double()
        num += ((d_ref - d_dp) ** 2).sum().item()
        den += (d_ref ** 2).sum().item()
    print("parameter displacement after 2 steps, data-parallel vs whole batch, relative L2:", (num / den) ** 0.5)
    assert den > 0 and (num / den) ** 0.5 < 5e-2


# ------------------------------------------------------------------ narrow-precision modes (BASELINE configs 2 / 5)
def _psnr(ref, x):
    mse = ((ref.double() - x.double()) ** 2).mean().item()
    return 10.0 * np.log10(float(ref.max().item()) ** 2 / max(mse, 1e-30))


@pytest.mark.parametrize("mode,bar_conv,bar_wgrad", [("bf16x2", 3e-5, 3e-5), ("bf16", 6e-3, 6e-3)])
def test_conv_precision_modes_layers(S, mode, bar_conv, bar_wgrad):
    """The two- and one-part forms of the matrix-core convolution / weight gradient against float64: two bf16 parts carry
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


@pytest.mark.parametrize("mode,bar_db", [("bf16x2", 70.0), ("bf16", 30.0)])
def test_conv_precision_modes_e2e_psnr(S, mode, bar_db):
    """The 12-cascade network at 320 x 320 with the convolutions in a narrow-precision mode, judged by PSNR against the
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


# ------------------------------------------------------------------ hipGraph capture of the training step
def test_captured_update_matches_eager(S):
    """CSModel.capture_update(): three replays of the captured 'Rec' step (two streams forked / joined inside the graph,
    AdamW step count in device memory, weights re-packed by the captured batch launch) leave bit-identical parameters and
    BatchNorm buffers to three eager steps from the same state."""
    n, c, h, w = 2, 3, 48, 80

    def make():
        cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                            weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=18,
                            sens_chans=8, pools=2, sens_pools=2)
        net = S.model.CSModel(cfg)
        net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
        _load(S, net.net_T, 41)
        _load(S, net.net_R, 42)
        net.to(DEV).train()
        for o in (net.optim_R, net.optim_T):
            o.device_step = True
        return net

    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    xf, xa = g(img_full), g(img_aux)
    eager = make()
    for _ in range(3):
        eager.set_input(xf, xa)
        eager.update()
    torch.cuda.synchronize()
    want = {k: v.detach().cpu().clone() for m in (eager.net_R, eager.net_T) for k, v in m.state_dict().items()}
    assert eager.optim_R.steps_taken() == 3
    cap = make()
    # 2 warm-up steps, undone again (restore=True, round 3: capturing must not train on duplicated data); the capture
    # itself does not execute
    graph = cap.capture_update(xf, xa, warmup=2)
    assert graph.mode.startswith("single-graph") and cap.optim_R.steps_taken() == 0
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert cap.optim_R.steps_taken() == 3 and cap.optim_T.steps_taken() == 3
    got = {k: v.detach().cpu() for m in (cap.net_R, cap.net_T) for k, v in m.state_dict().items()}
    bad = [k for k in want if not torch.equal(want[k], got[k])]
    assert not bad, bad[:5]
    # new data through the same graph: refill the captured input tensors in place
    f2, a2 = S.synth.phantom_pair(n, c, h, w, seed=77)
    xf.copy_(g(f2))
    xa.copy_(g(a2))
    graph.replay()
    eager.set_input(xf, xa)
    eager.update()
    torch.cuda.synchronize()
    assert all(torch.equal(p.cpu(), q.cpu()) for p, q in zip(eager.net_R.parameters(), cap.net_R.parameters()))
    # a learning-rate change between replays: lr lives in device memory next to the step count (FusedAdamW.sync_hyper)
    for net_ in (eager, cap):
        for o in (net_.optim_R, net_.optim_T):
            o.param_groups[0]["lr"] = 3e-5
            o.sync_hyper()
    graph.replay()
    eager.set_input(xf, xa)
    eager.update()
    torch.cuda.synchronize()
    assert all(torch.equal(p.cpu(), q.cpu()) for m1, m2 in ((eager.net_R, cap.net_R), (eager.net_T, cap.net_T))
               for p, q in zip(m1.parameters(), m2.parameters()))


def test_mixed_backward_precision_full_320(S):
    """cfg.bwd_dtype = 'bf16x2' (fp32-equivalent forward, backward convolutions on two bf16 parts) on the damped full-size
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
        wn64, name64, pe64 = _digest_errors(S, named, gold, f"{tag}.f64.grad.{nt}.")
        print(f"mixed backward, net_{nt}: probe-estimated relative L2 vs ref64 {pe64:.2e}, worst per-tensor norm {wn64:.2e} "
              f"({name64}); reference fp32-vs-fp64 {floor:.2e}")
        assert pe64 < max(3.0 * floor, 2e-3) and wn64 < max(3.0 * floor, 2e-3)


# ------------------------------------------------------------------ fp16 two-part operand format (default fp32-equivalent mode)
@pytest.mark.parametrize("scale", [1.0, 3e-7, 2e4])
@pytest.mark.parametrize("n,cin,cout,h,w,ks", [(2, 72, 36, 40, 40, 3), (1, 18, 18, 64, 64, 3), (2, 64, 64, 16, 32, 1), (1, 288, 144, 20, 20, 3)])
def test_f16x2_forward_and_gradients_vs_float64(S, n, cin, cout, h, w, ks, scale):
    """The fp16 two-part forms against float64: forward convolution (activations, no scale needed), data gradient and weight
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


@pytest.mark.parametrize("n,c,h,w,coff", [(2, 5, 320, 320, 0), (2, 6, 160, 160, 1), (1, 3, 40, 24, 0), (3, 4, 6, 8, 2)])
def test_act_bwd_second_gradient_source(S, n, c, h, w, coff):
    """san_act_bwd_up_amax: InstanceNorm + LeakyReLU backward whose incoming gradient is g + 0.25 * nearest_up(g2) (the
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


# ------------------------------------------------------------------ fp8 forward convolutions (BASELINE config 5)
def _e4m3(t):
    """OCP e4m3 round-to-nearest-even of a float32 tensor (|t| <= 448), as float64."""
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).double()


def _w_scale(wt):
    import math
    return 2.0 ** (7 - math.floor(math.log2(float(wt.abs().max()))))


@pytest.mark.parametrize("n,cin,cout,h,w,ks,kind", [
    (2, 72, 36, 40, 40, 3, "conv"),        # FLAT tile, weights direct
    (2, 18, 18, 64, 96, 3, "conv"),        # 32 x 8 tiles, half-padded channel blocks, operand-swapped epilogue
    (1, 96, 160, 24, 40, 3, "conv"),       # five channel blocks
    (2, 288, 288, 20, 20, 3, "conv"),      # split-K
    (2, 64, 32, 32, 48, 1, "conv"),        # 1x1 form
    (2, 72, 36, 16, 24, 1, "tconv"),       # transposed 2x2 s2: pixel-shuffle epilogue
])
def test_fp8_forward_matches_quantised_float64(S, n, cin, cout, h, w, ks, kind):
    """The fp8 mode's forward convolutions against float64 arithmetic on the SAME quantised operands: activations x 8 and
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
    """fp8 staging with the lazy InstanceNorm affine + LeakyReLU in front of the conversion, and activations beyond the
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


def test_fp8_mode_e2e_psnr_and_train_step(S):
    """Config 5: the 12-cascade network at 320 x 320 with fp8 e4m3 forward convolutions (fp32 FFT / DC / norms / losses),
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


def test_deferred_weight_gradient_reductions_are_bit_identical(S):
    """san_wgrad_defer: inside wgrad_overlap the matrix-core weight gradients queue the fixed-order reduction of their partial
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


# ------------------------------------------------------------------ late round-2 entry points
def test_group_norm_backward_aux_and_partials_add(S):
    """SAN_NORM_GROUP_BWD: the NormUnet statistics launch also writes the two per-plane values its backward needs (guarded
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
    """A split-K convolution followed by InstanceNorm writes the lazy affine in its reduction pass
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
