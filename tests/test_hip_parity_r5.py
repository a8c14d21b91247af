"""Round-5 GPU tests (through the C ABI): the overlapped step against the serial step at the bench's full sizes, a failed
recording, hand-off batch sizes, ``warp(interp=True)``, the fixed-point image gradient of the sampler."""
import os
import warnings

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def S():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spatialalignmentnetwork_amd import ops, synth, model, basemodel, _lib, autograd, cross

    class NS:
        pass

    ns = NS()
    ns.ops, ns.synth, ns.model, ns.base, ns.lib, ns.autograd, ns.cross = ops, synth, model, basemodel, _lib, autograd, cross
    return ns


def g(t):
    return t.to(DEV).contiguous()


def _fill(S, m, seed, damp=1.0):
    m.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=seed, damp=damp))


def _model(S, w, c, sparsity=0.25, **kw):
    cfg = S.base.Config(sparsity=sparsity, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                        weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, **kw)
    net = S.model.CSModel(cfg)
    net.net_mask.pruned = S.synth.equispaced_pruned(w, sparsity, 0)
    _fill(S, net.net_T, 41, damp=0.1)
    _fill(S, net.net_R, 42, damp=0.1)
    return net.to(DEV)


def _state(net):
    return {f"{s_}.{k}": v.detach().cpu().clone() for s_ in ("net_R", "net_T") for k, v in getattr(net, s_).state_dict().items()}


# ---------------------------------------------------------------------- overlap on / off at the sizes the bench and config 4 run
@pytest.mark.parametrize("tag,n,c,h,w,sparsity,steps", [("bench_n8_320", 8, 1, 320, 320, 0.25, 50),
                                                        ("config4_15x640x368", 1, 15, 640, 368, 0.125, 10)])
def test_overlapped_steps_equal_serial_steps_at_full_size(S, tag, n, c, h, w, sparsity, steps):
    """the default step runs three streams (weight gradients on the side stream, the sensitivity network
    beside the alignment network); with every overlap switched off the same kernels run one after the other.  ``steps``
    optimisation steps of the 12-cascade model (new data every step; eager, eager, then replays of the auto-recorded step) must
    leave BIT-identical parameters, BatchNorm buffers, reconstructions and losses in both forms.  The library carries no
    packed-fp32 instruction any more (tests/test_abi.py), which is what made two co-resident kernels disagree in round 4."""
    batches = [tuple(g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=700 + i)) for i in range(4)]

    def loop(overlap):
        S.ops.WGRAD_OVERLAP[0] = overlap
        S.model.SENS_OVERLAP[0] = overlap
        try:
            net = _model(S, w, c, sparsity=sparsity, num_cascades=12).train()
            sims, modes = [], []
            for it in range(steps):
                net.set_input(*batches[it % len(batches)])
                net.update()
                sims.append(net.loss_sim.detach().clone())
                modes.append(net.step_mode)
            torch.cuda.synchronize()
            return _state(net), torch.stack(sims).cpu(), net.img_rec.detach().cpu().clone(), modes
        finally:
            S.ops.WGRAD_OVERLAP[0] = True
            S.model.SENS_OVERLAP[0] = S.model.SENS_OVERLAP_DEFAULT

    st_s, sims_s, rec_s, modes_s = loop(False)
    torch.cuda.empty_cache()
    st_o, sims_o, rec_o, modes_o = loop(True)
    assert modes_o[0] == "eager" and modes_o[-1].startswith("replay"), modes_o
    assert torch.isfinite(sims_o).all() and torch.isfinite(sims_s).all(), (sims_s, sims_o)
    assert torch.equal(sims_s, sims_o), (tag, (sims_s - sims_o).abs().max().item())
    bad = [k for k in st_s if not torch.equal(st_s[k], st_o[k])]
    assert not bad, (tag, len(bad), bad[:5])
    assert torch.equal(rec_s, rec_o)


# --------------------------------------------------------------------------------------- a recording that fails
def test_failed_recording_leaves_the_model_where_the_eager_loop_would_be(S):
    """``update()`` tries to record the third step.  When the recording raises (here: after the warm-up step and the recorded step
    have both run and changed the weights), the model must be put back before the call falls back to the eager step -- otherwise
    the same batch gets three optimiser steps.  Four steps with a failing recorder == four eager steps, bit for bit."""
    n, c, h, w = 2, 3, 48, 80
    batches = [tuple(g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=300 + i)) for i in range(4)]

    def loop(break_recorder):
        net = _model(S, w, c, num_cascades=2, chans=6, sens_chans=4, pools=2, sens_pools=2).train()
        if break_recorder:
            real = net._record

            def failing(run, what, timer):
                real(run, what, timer)              # the step runs under the recorder (and trains) ...
                raise RuntimeError("stray operation (test)")      # ... and then the recording is refused

            net._record = failing
            net.memo_init.add("_record")
        else:
            net.auto_record = False
        modes = []
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            for b in batches:
                net.set_input(*b)
                net.update()
                modes.append(net.step_mode)
        torch.cuda.synchronize()
        return _state(net), modes, [str(x.message) for x in wlist], net.optim_R.steps_taken()

    want, _, _, steps_e = loop(False)
    got, modes, msgs, steps_f = loop(True)
    assert all(m == "eager" for m in modes), modes
    assert any("staying eager" in m for m in msgs), msgs
    assert steps_e == steps_f == len(batches)
    bad = [k for k in want if not torch.equal(want[k], got[k])]
    assert not bad, bad[:5]


# ------------------------------------------------------------------------------------------------ hand-off batches
def test_weight_gradient_handoff_batch_size_changes_nothing(S):
    """Queued weight gradients snapshot their operands (ops._on_side_stream): batches of 1, 4 and 16 launches behind one event give
    the same bits."""
    n, c, h, w = 2, 3, 48, 80
    batch = tuple(g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=77))

    def run(k):
        old = S.ops.WGRAD_BATCH[0]
        S.ops.WGRAD_BATCH[0] = k
        try:
            net = _model(S, w, c, num_cascades=2, chans=6, sens_chans=4, pools=2, sens_pools=2).train()
            net.auto_record = False
            for _ in range(2):
                net.set_input(*batch)
                net.update()
            torch.cuda.synchronize()
            return _state(net)
        finally:
            S.ops.WGRAD_BATCH[0] = old

    a, b, c_ = run(1), run(4), run(16)
    assert not [k for k in a if not torch.equal(a[k], b[k])]
    assert not [k for k in a if not torch.equal(a[k], c_[k])]


# ------------------------------------------------------------------------------------------------------------- warp(interp=True)
@pytest.mark.parametrize("hg,wg", [(24, 40), (48, 80), (96, 50)])
def test_warp_interp_resizes_like_the_reference(S, hg, wg):
    """cross.py:32-38: ``grid_sample`` on a grid of another size, then ``F.interpolate(size=img.shape[2:])`` (nearest) when
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
    """d/d img of the bilinear sampler: 64-bit fixed-point integer atomics instead of float atomics -- 20 launches give identical
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


# --------------------------------------------------------------------------------- bench.py through the launcher on one-rank RCCL
def test_bench_launcher_runs_the_exchange_on_rccl_with_one_rank():
    """``python bench.py --gpus 1`` with SAN_DIST_SINGLE=1 brings up a one-rank RCCL process group and runs the
    WHOLE data-parallel step through it -- communicator, probe all-reduce, per-cascade slices on the communication stream inside
    the recorded replays, the join in front of AdamW.  The line must say so (backend nccl, measured all-reduce time, the replayed
    step), and three steps must leave the parameters BIT-identical to the plain single-GPU run of the same command."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--main-only", "--no-kernel-timer", "--digest"]

    def run(single, mode=None):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("SAN_DIST_SINGLE", None)
        env.pop("SAN_GRAD_EXCHANGE", None)
        if single:
            env["SAN_DIST_SINGLE"] = "1"
        if mode:
            env["SAN_GRAD_EXCHANGE"] = mode
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    plain, single = run(False), run(True)
    assert plain["config"]["collective_backend"] is None and plain["allreduce_ms"] is None
    assert single["config"]["collective_backend"] == "nccl" and single["config"]["nccl_ranks"] == 1
    assert single["allreduce_ms"] is not None and single["allreduce_ms"] >= 0.0
    assert single["config"]["step_mode"].startswith("CSModel.update(): replay")
    assert single["n_gpus"] == 1 and single["steps"] == 3 and single["value"] > 0
    assert single["config"]["native_rccl"] is True, single["config"]["native_rccl_note"]     # the all-reduces are C-ABI tape entries
    assert single["optimizer_steps"] == plain["optimizer_steps"] >= 4
    assert single["state_digest"] == plain["state_digest"]
    # round 6: the line explains the exchange per rank (collectives' time, what the main stream waited for at the join, the rest hidden)
    ex = single["exchange"]
    assert ex["mode"] == "allreduce" and ex["slices_per_step"] == 14           # 12 cascades + the sensitivity net + net_T's buffer
    assert len(ex["collective_ms_per_rank"]) == len(ex["exposed_ms_per_rank"]) == len(ex["hidden_ms_per_rank"]) == 1
    assert 0.0 <= ex["exposed_ms_per_rank"][0] and ex["collective_ms_per_rank"][0] == pytest.approx(single["allreduce_ms"])
    # the reduce-scatter + all-gather form of the same sums (SAN_GRAD_EXCHANGE=rs_ag: ncclReduceScatter / ncclAllGather entry points
    # on the package's communicator): with one rank both are copies in place, so the parameters must again be bit-identical
    rsag = run(True, "rs_ag")
    assert rsag["exchange"]["mode"] == "rs_ag" and rsag["config"]["native_rccl"] is True
    assert rsag["state_digest"] == plain["state_digest"] and rsag["optimizer_steps"] == plain["optimizer_steps"]


# ----------------------------------------------------------------- round-5 kernels: direct small-channel convolution, one-stage GEMM
def _act64(x, sc, sh, slope):
    xd = x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
    return torch.where(xd >= 0, xd, xd * slope)


def _merge_stats(part):
    """(mean, biased variance, count) per (sample, channel) from statistics records [n, c, tiles, 3] = (count, mean, M2)."""
    cnt, mean, m2 = part[..., 0].double(), part[..., 1].double(), part[..., 2].double()
    assert torch.isfinite(part).all()
    tot = cnt.sum(-1)
    mu = (cnt * mean).sum(-1) / tot
    var = (m2 + cnt * (mean - mu[..., None]) ** 2).sum(-1) / tot
    return mu, var, tot


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
    """conv_direct_kernel (csrc/san_conv_mfma.hip) behind san_conv2d_fwd: outputs and merged statistics against float64 (3e-6, the
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
    """gemm1x1_f16_kernel (csrc/san_conv1x1.hip): ConvTranspose2d 2x2 s2 (varnet.py:159-192) forward with its statistics, and its
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
    """The plain 1x1 form (unet.py's 1x1 layers): channel views on both sides, bias, statistics; HW % 4 != 0 takes the scalar stores."""
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
    """Narrow-precision mode (one bf16 part): the GEMM form of a transposed convolution and of its data gradient against the tiled
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
    """100 launches each of the direct kernel (4 -> 18, 3x3, statistics) and of the GEMM form (transposed convolution 72 -> 36 with
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
