"""Round-3 GPU parity tests (through the C ABI): the LNCC backward, the autograd-visible entry points (the reference's
``loss.backward()`` idiom, varnet.py:559-560 / model.py:203-214) against the direct ``CSModel.update()`` chain and the
reference's own gradients, per-model arenas, and the device-normalisation fix of the gradient-maximum pool.
Tolerances are written next to each assertion together with what was measured."""
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
    from spatialalignmentnetwork_amd import (ops, synth, varnet, cross, unet, signal_utils, ssimloss, lnccloss, masks, model,
                                             basemodel, autograd)
    from oracle import cpu_ref as O

    class NS:
        pass

    ns = NS()
    ns.ops, ns.synth, ns.varnet, ns.cross, ns.unet, ns.sig, ns.ssim, ns.lncc = ops, synth, varnet, cross, unet, signal_utils, ssimloss, lnccloss
    ns.masks, ns.model, ns.base, ns.O, ns.autograd = masks, model, basemodel, O, autograd
    return ns


def g(t):
    return t.to(DEV).contiguous()


def _fill(S, m, seed, damp=1.0):
    m.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=seed, damp=damp))


def _pair():
    a = philox("loss.a", (2, 1, 40, 56), lo=0.0, hi=1.0)
    b = (a + 0.1 * philox("loss.b", (2, 1, 40, 56))).clamp(0, 1)
    return a, b


# ------------------------------------------------------------------------------------------- losses: backward kernels
def test_loss_backward_vs_reference_autograd(S):
    """lncc_loss / ms_lncc_loss / ssimloss .backward() (san_lncc_loss_bwd, san_smooth_pool_bwd, san_ssim_loss_bwd_dev) against
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
    """The bench batch (N = 8, 320 x 320): LNCC gradients vs oracle autograd (float64), plus symmetry of the kernel pair."""
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
    """lncc_loss(fixed, net_T.warp(moving, grid)).backward() -> d / d offset incl. the smoothness term, and
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


# ------------------------------------------------------------------------------------------- the reference's smoke idiom
@pytest.mark.parametrize("tag,shape", [("32", (2, 1, 32, 32)), ("48x80c3", (2, 3, 48, 80))])
def test_reference_smoke_idiom_matches_direct_chain(S, tag, shape):
    """``result = varnet(...); ssimloss(result, target).backward()`` (the reference's own smoke block, varnet.py:546-560):
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


def _rec_model(S, w, c, seed_T=41, seed_R=42, **kw):
    from spatialalignmentnetwork_amd.basemodel import Config
    from spatialalignmentnetwork_amd.model import CSModel
    cfg = Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                 weight_gan=0.0, weight_gan_sim=0.0, weight_sim=kw.pop("weight_sim", 1.0), use_amp=False, num_cascades=2,
                 chans=4, sens_chans=2, pools=2, sens_pools=2, **kw)
    net = CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    _fill(S, net.net_T, seed_T)
    _fill(S, net.net_R, seed_R)
    return net


def _grads(net):
    return [p.grad.detach().clone() for m in (net.net_R, net.net_T) for p in m.parameters()]


@pytest.mark.parametrize("shape,weight_sim", [((2, 1, 32, 32), 1.0), ((2, 3, 48, 80), 0.37)])
def test_loss_all_backward_matches_update_chain_bitwise(S, shape, weight_sim):
    """CSModel 'Rec': forwardT(); forwardR(); loss_all.backward() (the reference's model.py:203-214 idiom, through
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
    """Three optimisation steps written the reference's way (zero_grad; loss_all.backward(); optim.step()) give
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
    """Regime 'None' (model.py:195-204): forwardT under no_grad, only net_R trains; loss_all.backward() leaves net_T's
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


# ------------------------------------------------------------------------------------------- arenas / pools
def test_two_models_interleaved_do_not_share_tapes(S):
    """A.forward, B.forward, A.backward (a validation copy or an EMA next to the trained model): every
    model owns its arena, so A's gradients are bit-identical to the un-interleaved run."""
    n, c, h, w = 2, 1, 32, 32
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    img_b, aux_b = S.synth.phantom_pair(n, c, h, w, seed=77)

    def run(interleave: bool):
        A = _rec_model(S, w, c).to(DEV).train()
        B = _rec_model(S, w, c, seed_T=51, seed_R=52).to(DEV).train()
        A.set_input(g(img_full), g(img_aux))
        A.loss_all = 0
        A.forwardT()
        A.forwardR()
        if interleave:
            B.set_input(g(img_b), g(aux_b))
            B.loss_all = 0
            B.forwardT()
            B.forwardR()
        A.optim_R.zero_grad()
        A.optim_T.zero_grad()
        A.backward(train_T=True)
        torch.cuda.synchronize()
        return _grads(A), A.img_rec.detach().clone()

    (g0, r0), (g1, r1) = run(False), run(True)
    assert torch.equal(r0, r1)
    assert all(torch.equal(x, y) for x, y in zip(g0, g1)), "model B's forward changed model A's backward"
    # stand-alone modules own arenas too
    va, vb = S.varnet.VarNet(2, 2, 2, 4, 2, use_ref=False).to(DEV).train(), S.varnet.VarNet(2, 2, 2, 4, 2, use_ref=False).to(DEV).train()
    _fill(S, va, 42)
    _fill(S, vb, 43)
    k = g(cplx("arena.k", (2, 1, 32, 32)))
    mask = (~S.synth.equispaced_pruned(32, 0.25, 0)).to(DEV)
    gimg = g(philox("arena.g", (2, 1, 32, 32)))
    va(k, mask, None, 2)
    va.backward(gimg)
    want = [p.grad.clone() for p in va.parameters()]
    for p in va.parameters():
        p.grad.zero_()
    va(k, mask, None, 2)
    vb(k * 0.5, mask, None, 2)
    va.backward(gimg)
    assert all(torch.equal(p.grad, t) for p, t in zip(va.parameters(), want))


def test_vis_images_survive_the_next_step(S):
    """get_vis('images') hands out tensors the next step does not overwrite (the reference returns fresh tensors)."""
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


def test_amax_pool_resets_on_an_unindexed_device(S):
    """net.to(torch.device('cuda')) -- what the reference's train.py / eval.py do -- must still zero
    the gradient-maximum records every step ('cuda' == 'cuda:0').  A poisoned record (exponent 250) would otherwise scale
    every later gradient to zero."""
    ops = S.ops
    ops.AMAX.reset(torch.device("cuda"))
    rec = ops.AMAX.next(torch.device("cuda"))
    rec.fill_(0x7F000000)                                   # a huge recorded maximum
    assert ops.AMAX.idx == 1
    ops.AMAX.reset(torch.device("cuda"))                   # unindexed
    assert ops.AMAX.idx == 0 and int(rec.abs().max().item()) == 0
    rec = ops.AMAX.next(DEV)
    rec.fill_(0x7F000000)
    ops.AMAX.reset("cuda")
    assert int(rec.abs().max().item()) == 0
    # and a whole model moved with the unindexed device trains to the same bits as one moved to cuda:0
    n, c, h, w = 2, 1, 32, 32
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    outs = []
    for dev in (torch.device("cuda"), torch.device(DEV)):
        net = _rec_model(S, w, c).to(dev).train()
        for _ in range(2):
            net.set_input(img_full.to(dev), img_aux.to(dev))
            net.update()
        torch.cuda.synchronize()
        outs.append([p.detach().clone() for m in (net.net_R, net.net_T) for p in m.parameters()])
    assert all(torch.equal(x, y) for x, y in zip(*outs))


def test_alignment_backward_in_eval_mode_vs_oracle_autograd(S):
    """Backward through the alignment network in eval mode (BatchNorm on running statistics): every
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


# ------------------------------------------------------------------------------------------- data parallel: captured step
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker3(rank, world, port, path, captured):
    """Two ranks sharing cuda:0 over gloo run three 'Rec' steps, eagerly or as a captured step (capture_update under a
    process group: gloo cannot be captured, so the step is two graphs around the eager exchange)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from spatialalignmentnetwork_amd import dist as sdist, synth
    from spatialalignmentnetwork_amd.basemodel import Config
    from spatialalignmentnetwork_amd.model import CSModel
    d = sdist.init("gloo")
    h, w = 32, 32
    torch.manual_seed(100 + rank)
    cfg = Config(sparsity=0.25, lr=1e-4, shape=w, coils=1, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                 weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=4,
                 sens_chans=2, pools=2, sens_pools=2)
    net = CSModel(cfg)
    net.net_mask.pruned = synth.equispaced_pruned(w, 0.25, 0)
    if rank == 0:
        for sub, sd in (("net_T", 41), ("net_R", 42)):
            m = getattr(net, sub)
            m.load_state_dict(synth.fill_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=sd))
    net.to("cuda:0").train()
    img_full, img_aux = synth.phantom_pair(4, 1, h, w, seed=40)
    lo, hi = sdist.shard_bounds(4, rank, world)
    xf, xa = img_full[lo:hi].to("cuda:0").contiguous(), img_aux[lo:hi].to("cuda:0").contiguous()
    mode = "eager"
    if captured == 2:
        step = net.record_update(xf, xa, warmup=1)
        mode = step.mode
        for _ in range(3):
            step.replay()
    elif captured:
        step = net.capture_update(xf, xa, warmup=1)
        mode = step.mode
        for _ in range(3):
            step.replay()
    else:
        net.sync_replicas()                                 # before the first set_input
        for _ in range(3):
            net.set_input(xf, xa)
            net.update()
    torch.cuda.synchronize()
    out = {"mode": mode, "steps": net.optim_R.steps_taken(),
           "params": {f"{s_}.{k}": v.cpu() for s_ in ("net_R", "net_T") for k, v in getattr(net, s_).state_dict().items()}}
    torch.save(out, f"{path}/rank{rank}_{int(captured)}.pt")
    d.barrier()
    d.destroy_process_group()


def test_captured_step_under_a_process_group_two_ranks(S, tmp_path):
    """VERDICT r2 #4a: capture_update works with an active process group.  Two ranks (gloo, one GPU): the captured step
    (two graphs around the exchange, since gloo stages through the host; with RCCL the all-reduce is captured inside one
    graph) leaves bit-identical parameters to the eager data-parallel steps, on both ranks, and capturing itself does not
    advance the optimiser."""
    import torch.multiprocessing as mp
    for captured in (0, 1, 2):
        mp.spawn(_dp_worker3, args=(2, _free_port(), str(tmp_path), captured), nprocs=2, join=True)
    e0, e1 = torch.load(tmp_path / "rank0_0.pt"), torch.load(tmp_path / "rank1_0.pt")
    c0, c1 = torch.load(tmp_path / "rank0_1.pt"), torch.load(tmp_path / "rank1_1.pt")
    assert c0["mode"].startswith("two graphs") and e0["mode"] == "eager"
    assert e0["steps"] == c0["steps"] == c1["steps"] == 3
    for k in e0["params"]:
        if "running_" not in k and "num_batches" not in k:      # BatchNorm statistics stay per replica (unet.py:125 semantics)
            assert torch.equal(e0["params"][k], e1["params"][k]), ("eager replicas diverged", k)
            assert torch.equal(c0["params"][k], c1["params"][k]), ("captured replicas diverged", k)
        assert torch.equal(e0["params"][k], c0["params"][k]), ("captured != eager on rank 0", k)
        assert torch.equal(e1["params"][k], c1["params"][k]), ("captured != eager on rank 1", k)
    # the recorded-step form (CSModel.record_update) under the same process group
    r0, r1 = torch.load(tmp_path / "rank0_2.pt"), torch.load(tmp_path / "rank1_2.pt")
    assert r0["mode"].startswith("recorded step") and r0["steps"] == r1["steps"] == 3
    for k in e0["params"]:
        assert torch.equal(e0["params"][k], r0["params"][k]), ("recorded != eager on rank 0", k)
        assert torch.equal(e1["params"][k], r1["params"][k]), ("recorded != eager on rank 1", k)


# ------------------------------------------------------------------------------------------- the bench batch (N = 8)
def _probe_idx(S, name, numel, k=16):
    return S.synth._rng("probe." + name, 0).integers(0, numel, k)


def _digest_errors(S, named_grads, gold, pre):
    """Per-network relative L2 (from per-tensor norms and 16 probes per tensor) of our gradients against a digest fixture."""
    names = [str(s_) for s_ in gold[pre + "names"]]
    l2 = gold[pre + "l2"]
    grads = dict(named_grads)
    worst_norm, worst_name, num, den = 0.0, "", 0.0, 0.0
    for i, nm in enumerate(names):
        got = grads[nm].detach().double().reshape(-1).cpu()
        assert got.numel() == int(gold[pre + "numel"][i]), nm
        e = abs(got.norm().item() - float(l2[i])) / max(float(l2[i]), 1e-30)
        if float(l2[i]) > 1e-3 * float(l2.max()) and e > worst_norm:
            worst_norm, worst_name = e, nm
        pr = got[torch.from_numpy(_probe_idx(S, nm, got.numel()))]
        want = torch.from_numpy(gold[pre + "probes"][i])
        num += ((pr - want) ** 2).sum().item() * got.numel() / 16.0
        den += float(l2[i]) ** 2
    return worst_norm, worst_name, (num / den) ** 0.5


def test_train_step_bench_batch_n8_golden(S):
    """the batch the bench is quoted on -- one 'Rec' training step at N = 8, 320 x 320, 12 cascades, chans 18
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
        wn64, name64, pe64 = _digest_errors(S, named, gold, f"f64.grad.{nt}.")
        wn32, name32, pe32 = _digest_errors(S, named, gold, f"f32.grad.{nt}.")
        print(f"[n8] net_{nt}: per-tensor norm error vs ref64 {wn64:.2e} ({name64}), vs ref32 {wn32:.2e}; probe-estimated relative "
              f"L2 vs ref64 {pe64:.2e}, vs ref32 {pe32:.2e}; reference fp32-vs-fp64: exact {floor:.2e}, probe-estimated "
              f"{floor_probe:.2e}, worst per-tensor norm {floor_norm:.2e}")
        assert pe64 < max(3.0 * max(floor, floor_probe), 2e-3), (nt, pe64, floor, floor_probe)
        assert wn64 < max(3.0 * max(floor, floor_norm), 2e-3), (nt, wn64, name64, floor, floor_norm)
    for k_ in gold.files:
        if k_.startswith("f32.bn_after.T."):
            got = dict(net.net_T.named_buffers())[k_[len("f32.bn_after.T."):]]
            assert torch.allclose(got.cpu(), as_t(gold[k_]), rtol=2e-4, atol=2e-6), k_


def _psnr(ref, x):
    mse = ((ref.double() - x.double()) ** 2).mean().item()
    return 10.0 * np.log10(float(ref.max().item()) ** 2 / max(mse, 1e-30))


def test_narrow_precision_psnr_on_trained_like_weights(S):
    """bf16 / fp8 convolutions judged where the judgement means something -- on the 'damped' weight set
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


@pytest.mark.parametrize("n,c,h,w", [(2, 5, 16, 24), (1, 3, 320, 320), (2, 4, 40, 40)])
def test_act_bwd_destination_modes_are_bit_identical(S, n, c, h, w):
    """san_act_bwd_ex_amax: the pixel-unshuffled store equals san_act_bwd_amax + san_unshuffle2_fwd bit for bit (one-pass plane
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


# ------------------------------------------------------------------------------------------- recorded step (host-light replay)
def test_recorded_step_replays_bit_identically(S):
    """CSModel.record_update(): one recorded 'Rec' step replayed three times (a flat loop over the recorded C-ABI calls,
    stream / event operations and torch operations) leaves bit-identical parameters and BatchNorm buffers to three eager
    steps from the same state; recording itself does not advance the model; new data goes through the static input tensors;
    a learning-rate change is followed; an eager step afterwards continues correctly."""
    n, c, h, w = 2, 3, 48, 80

    def make():
        cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                            weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=18,
                            sens_chans=8, pools=2, sens_pools=2)
        net = S.model.CSModel(cfg)
        net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
        _fill(S, net.net_T, 41)
        _fill(S, net.net_R, 42)
        net.to(DEV).train()
        for o in (net.optim_R, net.optim_T):
            o.device_step = True
        return net

    def state(net):
        return {f"{s_}.{k}": v.detach().cpu().clone() for s_ in ("net_R", "net_T") for k, v in getattr(net, s_).state_dict().items()}

    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    xf, xa = g(img_full), g(img_aux)
    eager = make()
    for _ in range(3):
        eager.set_input(xf, xa)
        eager.update()
    torch.cuda.synchronize()
    want = state(eager)
    cap = make()
    before = state(cap)
    step = cap.record_update(xf, xa, warmup=2)
    assert step.mode.startswith("recorded step") and cap.optim_R.steps_taken() == 0
    after = state(cap)
    assert all(torch.equal(before[k], after[k]) for k in before), "recording advanced the model"
    for _ in range(3):
        step.replay()
    torch.cuda.synchronize()
    assert cap.optim_R.steps_taken() == 3 and cap.optim_T.steps_taken() == 3
    got = state(cap)
    bad = [k for k in want if not torch.equal(want[k], got[k])]
    assert not bad, bad[:5]
    assert torch.equal(cap.img_rec, eager.img_rec) and torch.equal(cap.loss_sim, eager.loss_sim)
    # new data through the static inputs, and a learning-rate change
    f2, a2 = S.synth.phantom_pair(n, c, h, w, seed=77)
    xf.copy_(g(f2))
    xa.copy_(g(a2))
    for net_ in (eager, cap):
        for o in (net_.optim_R, net_.optim_T):
            o.param_groups[0]["lr"] = 3e-5
            o.sync_hyper()
    step.replay()
    eager.set_input(xf, xa)
    eager.update()
    torch.cuda.synchronize()
    want, got = state(eager), state(cap)
    assert all(torch.equal(want[k], got[k]) for k in want)
    # back to eager launching on the recorded model
    for net_ in (eager, cap):
        net_.set_input(xf, xa)
        net_.update()
    torch.cuda.synchronize()
    want, got = state(eager), state(cap)
    assert all(torch.equal(want[k], got[k]) for k in want)


def test_recorded_forward_pass_replays_bit_identically(S):
    """CSModel.record_forward(): the inference pass as a recorded step; replays on new data through the static inputs give the
    bit-identical reconstruction / warp / loss of the eager pass."""
    n, c, h, w = 2, 1, 64, 64
    net = _rec_model(S, w, c).to(DEV).eval()
    img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
    xf, xa = g(img_full), g(img_aux)
    rec = net.record_forward(xf, xa)
    f2, a2 = S.synth.phantom_pair(n, c, h, w, seed=77)
    xf.copy_(g(f2))
    xa.copy_(g(a2))
    rec.replay()
    torch.cuda.synchronize()
    got = {k: getattr(net, k).detach().clone() for k in ("img_rec", "img_warped", "loss_sim", "loss_smooth", "img_sampled_rss")}
    ref = _rec_model(S, w, c).to(DEV).eval()
    with torch.no_grad():
        ref.set_input(g(f2), g(a2))
        ref.loss_all = 0
        ref.forwardT()
        ref.forwardR()
    torch.cuda.synchronize()
    for k, v in got.items():
        assert torch.equal(v, getattr(ref, k)), k
