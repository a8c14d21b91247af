"""Round-4 GPU parity tests (through the C ABI): the reference's training loop on ``CSModel.update()`` with the auto-recorded step,
the checked replay, packed-weight freshness across replays, the persistent ("stream") convolution against float64 and against
the one-tile-per-workgroup kernel.  Tolerances are written next to each assertion together with what was measured."""
import os

import pytest
import torch

from conftest import philox

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def S():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spatialalignmentnetwork_amd import ops, synth, model, basemodel, _lib

    class NS:
        pass

    ns = NS()
    ns.ops, ns.synth, ns.model, ns.base, ns.lib = ops, synth, model, basemodel, _lib
    return ns


def g(t):
    return t.to(DEV).contiguous()


def _fill(S, m, seed, damp=1.0):
    m.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=seed, damp=damp))


def _model(S, w, c, reg="Rec", chans=18, **kw):
    cfg = S.base.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg=reg, mask="equispaced", weight_smooth=1000.0,
                        weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=chans,
                        sens_chans=8, pools=2, sens_pools=2, **kw)
    net = S.model.CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
    _fill(S, net.net_T, 41)
    _fill(S, net.net_R, 42)
    return net.to(DEV)


def _state(net):
    return {f"{s_}.{k}": v.detach().cpu().clone() for s_ in ("net_R", "net_T") for k, v in getattr(net, s_).state_dict().items()}


# ------------------------------------------------------------------------------------------- the reference's training loop
@pytest.mark.parametrize("reg", ["Rec", "None"])
def test_reference_train_loop_on_update_auto_records_bit_identically(S, reg):
    """/root/reference/train.py:212-217 verbatim: ``net.set_input(*batch); net.update()`` with a NEW batch every iteration.
    ``update()`` runs two steps eagerly, records the third (the recording does not advance the model) and replays from then
    on; a validation pass (``eval(); set_input; test(); train()``) in between, a learning-rate change and a change of the loss
    weight (which drops the recording) are followed.  Parameters, BatchNorm buffers, the reconstruction and the scalar losses
    are BIT-identical to the same loop with auto-recording switched off."""
    n, c, h, w = 2, 3, 48, 80
    batches = [tuple(g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=100 + i)) for i in range(8)]

    def loop(auto):
        net = _model(S, w, c, reg=reg)
        net.auto_record = auto
        modes, vis = [], None
        for it, batch in enumerate(batches):
            net.train()
            net.set_input(*batch)
            net.update()
            modes.append(net.step_mode)
            if it == 3:                                 # validation in between (eval.py / train.py:240-260)
                net.eval()
                net.set_input(*batches[0])
                psnr = net.test()
                vis = (psnr, net.img_rec.detach().clone())
            if it == 4:
                for o in (net.optim_R, net.optim_T):
                    o.param_groups[0]["lr"] = 3e-5
            if it == 5:
                net.cfg.weight_smooth = 500.0           # part of the recording's key: back to eager, re-recorded two steps later
        torch.cuda.synchronize()
        scal = net.get_vis("scalars")["scalars"]
        return net, modes, vis, scal

    ref, modes_e, vis_e, scal_e = loop(False)
    net, modes_a, vis_a, scal_a = loop(True)
    assert all(m == "eager" for m in modes_e)
    assert modes_a[0] == modes_a[1] == "eager" and all(m.startswith("replay") for m in modes_a[2:6]), modes_a
    assert modes_a[6] == "eager" and modes_a[7] == "eager", modes_a        # the new key has seen two steps only
    want, got = _state(ref), _state(net)
    bad = [k for k in want if not torch.equal(want[k], got[k])]
    assert not bad, bad[:5]
    assert torch.equal(ref.img_rec, net.img_rec) and torch.equal(ref.img_warped, net.img_warped)
    assert vis_e[0] == vis_a[0] and torch.equal(vis_e[1], vis_a[1])        # the eager validation pass saw the replayed weights
    assert scal_e == scal_a, (scal_e, scal_a)
    assert net.optim_R.steps_taken() == len(batches)


def test_replay_checks_return_codes(S):
    """A recorded C-ABI call that fails inside a replay raises (RecordedStep.replay dropped every return code)."""
    n, c, h, w = 1, 1, 32, 32
    net = _model(S, w, c, chans=4).train()
    xf, xa = (g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=3))
    step = net.record_update(xf, xa, warmup=1)
    step.replay()
    torch.cuda.synchronize()
    # corrupt one recorded call: a null pointer makes the entry point return SAN_E_ARG
    idx = next(i for i, (fn, args, kind) in enumerate(step.calls) if kind == 1 and getattr(fn, "__name__", "") == "san_norm_finalize")
    fn, args, kind = step.calls[idx]
    step.calls[idx] = (fn, (None,) + tuple(args[1:]), kind)
    step.invalidate()                                   # (the native tapes are built from `calls` at the first replay)
    with pytest.raises(RuntimeError, match="san_norm_finalize failed"):
        step.replay()
    torch.cuda.synchronize()


def test_recorded_forward_follows_weight_changes(S):
    """a recorded forward pass re-packs its weight images when the weights changed since its last replay
    (optimiser step, load_state_dict), and an eager forward after training replays sees the new weights."""
    n, c, h, w = 2, 1, 64, 64
    net = _model(S, w, c).eval()
    xf, xa = (g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=40))
    rec = net.record_forward(xf, xa)
    out_rec = net.img_rec                               # the recording's output tensor: every replay refreshes it in place
    rec.replay()
    torch.cuda.synchronize()
    first = out_rec.detach().clone()
    # a training step changes every weight (and re-points net.img_* at its own tensors)
    net.train()
    net.set_input(xf, xa)
    net.update()
    net.eval()
    rec.replay()
    torch.cuda.synchronize()
    got = out_rec.detach().clone()
    with torch.no_grad():
        net.set_input(xf, xa)
        net.loss_all = 0
        net.forwardT()
        net.forwardR()
    torch.cuda.synchronize()
    assert not torch.equal(first, got), "the step did not change the reconstruction"
    assert torch.equal(got, net.img_rec), "the replayed forward pass ran on stale packed weights"
    # recorded TRAINING replays followed by an eager forward pass
    net.train()
    step = net.record_update(xf, xa, warmup=1)
    with torch.no_grad():                               # an eager pass brings the pack registries up to date ...
        net.eval()
        net.set_input(xf, xa)
        net.loss_all = 0
        net.forwardT()
        net.forwardR()
    for _ in range(2):                                  # ... then the weights move under replays
        step.replay()
    torch.cuda.synchronize()
    ref = _model(S, w, c).eval()
    for s_ in ("net_T", "net_R"):
        getattr(ref, s_).load_state_dict(getattr(net, s_).state_dict())
    with torch.no_grad():
        for m in (net, ref):
            m.eval()
            m.set_input(xf, xa)
            m.loss_all = 0
            m.forwardT()
            m.forwardR()
    torch.cuda.synchronize()
    assert torch.equal(net.img_rec, ref.img_rec), "the eager pass after replays ran on stale packed weights"


# ------------------------------------------------------------------------------------------- stream convolution
def _conv_ref64(x, sc, sh, slope, wt, bias):
    a = x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
    a = torch.where(a >= 0, a, a * slope)
    return torch.nn.functional.conv2d(a, wt.double(), None if bias is None else bias.double(), padding=1)


@pytest.mark.parametrize("cin,cout,h,w,bias", [(18, 18, 160, 192, False), (36, 18, 160, 160, False), (18, 36, 160, 160, False),
                                               (36, 36, 168, 160, True), (72, 36, 160, 160, False), (96, 32, 160, 160, True),
                                               (20, 16, 160, 160, False), (48, 48, 160, 160, False)])
def test_stream_convolution_vs_float64_and_tile_kernel(S, cin, cout, h, w, bias):
    """conv3x3_stream_kernel (persistent workgroups; csrc/san_conv_stream.hip) on every template form, one to four 24-channel
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
    """The stream kernel as the data gradient of a 3x3 convolution: dy of magnitude 1e-7 scaled by its recorded power of two
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
    """200 launches of the persistent kernel on the same data: bit-identical outputs and statistics (and no hang: an early build
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
    """The persistent kernel's one-part form (san_set_conv_precision 'bf16': BASELINE configs[1] as written): activations and
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

# ------------------------------------------------------------------------------------------- bench batch, eval mode
def test_eval_bench_batch_n8_golden(S):
    """the bench batch in EVAL mode -- N = 8 slices of 320 x 320, 12 cascades, chans 18 -- against the
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


@pytest.mark.parametrize("tag", ["a", "b"])
def test_metric_ssim_matches_the_skimage_algorithm(S, tag):
    """metrics.py:40-43 averages skimage.metrics.structural_similarity(g[0], p[0], data_range=1) over the batch.  The fixture
    value is skimage's algorithm restated on scipy.ndimage.uniform_filter (make_golden.py::_ssim_skimage_algorithm; skimage is
    not in the image): metric_SSIM (= 1 - ssimloss on the device) must agree to 2e-5 (measured 3e-7)."""
    from conftest import load_golden
    from spatialalignmentnetwork_amd import metrics as M
    gold = load_golden("metrics.npz")
    gt, pred = torch.from_numpy(gold[f"{tag}.gt"]), torch.from_numpy(gold[f"{tag}.pred"])
    want = float(gold[f"{tag}.ssim_skimage_algorithm"])
    got = M.ssim(g(gt), g(pred))
    assert abs(got - want) < 2e-5, (got, want)


# ------------------------------------------------------------------------------------------- gradients on the SHIPPED kernels
@pytest.mark.parametrize("tag,shape", [("32", (2, 1, 32, 32)), ("48x80c3", (2, 3, 48, 80))])
def test_full_rec_step_gradients_elementwise_on_shipped_kernels(S, tag, shape):
    """EVERY parameter gradient of a 'Rec' step, element-wise, against the reference's own gradients
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


def test_test_metrics_single_sync_matches_the_separate_calls(S):
    """metrics.test_metrics (what CSModel.test() uses: ONE host synchronisation for MSE / MAE / PSNR / SSIM / MI) against the
    per-metric functions (one synchronisation each): identical values."""
    from spatialalignmentnetwork_amd import metrics as M
    gt = g(philox("tm.gt", (3, 1, 48, 64), lo=0.0, hi=1.0))
    pred = (gt + 0.05 * g(philox("tm.d", (3, 1, 48, 64)))).clamp(0, 1)
    warped = (gt + 0.2 * g(philox("tm.w", (3, 1, 48, 64)))).clamp(0, 1)
    m = M.test_metrics(gt, pred, warped)
    assert m["MSE"] == M.mse(gt, pred) and m["MAE"] == M.mae(gt, pred) and m["PSNR"] == M.psnr(gt, pred)
    assert m["MI"] == M.mi(gt, warped) and abs(m["SSIM"] - M.ssim(gt, pred)) < 1e-7


# ------------------------------------------------------------------------------------------- RCCL, one rank (SAN_DIST_SINGLE)
def _rccl_single_worker(rank, port, path, with_group):
    """Five update() calls (two eager, the recording, two replays) and one captured step of a small 'Rec' model; with_group: under a
    ONE-rank RCCL process group with SAN_DIST_SINGLE=1, i.e. with the whole gradient exchange of the data-parallel step."""
    import types
    if with_group:
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SAN_DIST_SINGLE="1")
    from spatialalignmentnetwork_amd import basemodel, dist as sdist, model as smodel, synth
    dev = torch.device(DEV)
    d = sdist.init("nccl", dev) if with_group else None
    S_ = types.SimpleNamespace(base=basemodel, model=smodel, synth=synth)
    n, c, h, w = 2, 1, 48, 80
    net = _model(S_, w, c, chans=4).train()
    net.time_exchange = with_group
    info = {"backend": sdist.backend() if with_group else None, "modes": []}
    for it in range(5):
        net.set_input(*(g(t) for t in synth.phantom_pair(n, c, h, w, seed=300 + it)))
        net.update()
        info["modes"].append(net.step_mode)
    torch.cuda.synchronize()
    info["slices"] = getattr(net, "exchange_slices", None)
    xf, xa = (g(t) for t in synth.phantom_pair(n, c, h, w, seed=310))
    cap = net.capture_update(xf, xa, warmup=1)
    cap.replay()
    torch.cuda.synchronize()
    info["capture_mode"] = cap.mode
    info["state"] = _state(net)
    torch.save(info, f"{path}/{'rccl' if with_group else 'plain'}.pt")
    if d is not None:
        d.destroy_process_group()


def test_gradient_exchange_runs_on_rccl_with_one_rank(S, tmp_path):
    """'RCCL code paths have literally never run.'  A one-GPU box cannot show the transport, but it can run every
    RCCL call site of the data-parallel step: dist.init("nccl") (communicator + probe all-reduce), the per-cascade slices of
    net_R's flat buffer launched in reverse order from inside VarNet.backward on the communication stream, net_T's buffer, the
    join in front of AdamW, the same collectives inside a recorded step's replays, and the eager exchange between the two graphs of a captured step.  With
    SAN_DIST_SINGLE=1 a one-rank group counts as data-parallel; sums over one rank change nothing, so parameters and BatchNorm
    buffers must equal the plain single-process run BIT for bit."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    mp.spawn(_rccl_single_worker, args=(port, str(tmp_path), True), nprocs=1, join=True)
    mp.spawn(_rccl_single_worker, args=(port, str(tmp_path), False), nprocs=1, join=True)
    a, b = torch.load(tmp_path / "rccl.pt"), torch.load(tmp_path / "plain.pt")
    assert a["backend"] == "nccl"
    assert a["modes"][0] == "eager" and a["modes"][-1].startswith("replay"), a["modes"]
    # net_R went out in slices: the cascades in reverse order, then the sensitivity net; net_T's whole buffer (None) last
    sl = a["slices"]
    assert sl is not None and len(sl) >= 4 and sl[-1] is None and all(r is not None for r in sl[:-1]), sl
    los = [r[0] for r in sl[:2]]
    assert los[0] > los[1], f"cascade slices not in reverse order: {sl}"
    assert a["capture_mode"] == "two graphs around an eager exchange", a["capture_mode"]
    assert all(torch.equal(a["state"][k], b["state"][k]) for k in b["state"]), "the one-rank exchange changed the step"


def test_recording_keeps_every_packed_image_it_rewrites(S):
    """A recorded step's weight-packing launch re-packs EVERY job of the table it was recorded with -- other live models' too.  When
    such a model is freed later and its jobs are pruned, the packed buffers must stay allocated for as long as the recording lives:
    otherwise a replay writes packed weights into memory the allocator has handed to somebody else."""
    import gc
    ops = S.ops
    n, c, h, w = 1, 1, 32, 32
    other = _model(S, w, c, chans=4).eval()
    xf, xa = (g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=5))
    with torch.no_grad():
        other.set_input(xf, xa)
        other.loss_all = 0
        other.forwardT()
        other.forwardR()                                    # `other`'s weights are registered with the pack registries
    net = _model(S, w, c, chans=4).eval()
    rec = net.record_forward(xf.clone(), xa.clone())
    ptrs = {j["packed"].data_ptr() for reg in (ops.PACKS, ops.PACKS16) for j in reg.order}
    assert ptrs, "nothing registered"
    kept = {t.data_ptr() for item in rec.keep if isinstance(item, list) for t in item if isinstance(t, torch.Tensor)}
    assert ptrs <= kept, "the recording does not hold every packed image its packing launch writes"
    del other
    gc.collect()
    for reg in (ops.PACKS, ops.PACKS16):
        reg._prune()                                        # `other`'s jobs are gone from the registries ...
    live = {j["packed"].data_ptr() for reg in (ops.PACKS, ops.PACKS16) for j in reg.jobs.values()}
    assert len(live) < len(ptrs)
    junk = [torch.full((1 << 18,), 7.0, device=DEV) for _ in range(32)]      # ... and the allocator is asked for fresh blocks
    rec.replay()
    torch.cuda.synchronize()
    assert all(bool((t == 7.0).all()) for t in junk), "a replay wrote into memory that no longer belongs to a packed image"


@pytest.mark.parametrize("c,h,w", [(1, 48, 80), (3, 80, 112)])
def test_no_kernel_writes_outside_its_arena_buffers(S, c, h, w):
    """Guard bands around every arena buffer (ops.ARENA_GUARD): a full training step -- odd plane sizes at the lower levels, the
    W % 4 != 0 forms, split-K scratch, rotating dy copies -- leaves all of them intact; a deliberate store past a buffer's end is
    reported.  (A store outside a buffer is harmless while the step's streams run one after the other and corrupts a neighbour
    once they overlap: the check the stream-overlap work of round 4 needed.)"""
    ops = S.ops
    before = len(ops._GUARDS)
    ops.ARENA_GUARD[0] = True
    try:
        net = _model(S, w, c, chans=4).train()
        xf, xa = (g(t) for t in S.synth.phantom_pair(1, c, h, w, seed=11))
        net.auto_record = False
        for _ in range(2):
            net.set_input(xf, xa)
            net.update()
        assert len(ops._GUARDS) > before + 20, "the step's arena buffers were not guarded"
        assert ops.arena_guard_report() == []
        key, raw, nbytes = ops._GUARDS[-1]
        raw[ops._GUARD_BYTES + nbytes + 3] = 0            # one byte past the buffer's end
        rep = ops.arena_guard_report()
        assert len(rep) == 1 and rep[0][1:] == ("back", 1, 3), rep
        raw[ops._GUARD_BYTES + nbytes + 3] = ops._GUARD_PATTERN
    finally:
        ops.ARENA_GUARD[0] = False
        del ops._GUARDS[before:]


def test_plane_activation_backward_is_bit_stable_beside_another_streams_convolutions(S):
    """The one-pass InstanceNorm backward (act_bwd_plane_kernel) on one stream while data-gradient convolutions run on another,
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
