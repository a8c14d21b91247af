"""GPU parity tests (through the C ABI), component: the step's runtime: recorded / captured steps, stream overlap, arenas, determinism, the fused optimiser (rows f1, d).
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


def test_fused_adamw_matches_torch(S):
    """[round 1] san_adamw_step over flat buffers == torch.optim.AdamW on the same tensors (4 steps, weight decay on and
    off, a 1/world gradient scale).  Tolerance 2e-6 relative: same fp32 formula, different operation order."""
    from spatialalignmentnetwork_amd.optim import FusedAdamW
    shapes = [(18, 3, 3, 3), (18,), (7, 5), (1,), (36, 18, 3, 3)]
    for wd, scale in ((0.0, 1.0), (0.01, 0.5)):
        ref = [torch.nn.Parameter(philox(f"ad.p{i}", sh).clone()) for i, sh in enumerate(shapes)]
        mine = [torch.nn.Parameter(g(r.detach().clone())) for r in ref]
        o_ref = torch.optim.AdamW(ref, lr=1e-2, weight_decay=wd)
        o_mine = FusedAdamW(mine, lr=1e-2, weight_decay=wd)
        for step in range(4):
            o_mine.zero_grad()
            for i, (r, m) in enumerate(zip(ref, mine)):
                gr = philox(f"ad.g{step}.{i}", tuple(r.shape))
                r.grad = gr.clone() * scale
                m.grad.copy_(g(gr))                       # p.grad is a view into the flat buffer
            o_ref.step()
            o_mine.step(grad_scale=scale)
        for r, m in zip(ref, mine):
            assert rel_err(m.detach().cpu(), r.detach()) < 2e-6
        # parameters are views of one flat buffer and survive as the same Parameter objects
        b = o_mine.bucket()
        assert all(b.flat_p.data_ptr() <= m.data_ptr() < b.flat_p.data_ptr() + 4 * b.total for m in mine)


@pytest.mark.gpu
def test_training_steps_are_bit_reproducible_and_overlap_changes_nothing(S):
    """[round 1] Three 'Rec' optimisation steps (48 x 80, 3 coils) run twice from the same state give bit-identical parameters
    (no float atomics anywhere: partial sums are added in fixed orders), and running the weight gradients on the side
    stream (ops.wgrad_overlap, the default) gives bit-identical parameters to running them in line."""
    from spatialalignmentnetwork_amd.basemodel import Config
    from spatialalignmentnetwork_amd.model import CSModel
    n, c, h, w = 2, 3, 48, 80

    def run(overlap: bool):
        S.ops.WGRAD_OVERLAP[0] = overlap
        try:
            cfg = Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                         weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=4,
                         sens_chans=2, pools=2, sens_pools=2)
            net = CSModel(cfg)
            net.net_mask.pruned = S.synth.equispaced_pruned(w, 0.25, 0)
            net.net_T.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net.net_T.state_dict().items()], seed=41))
            net.net_R.load_state_dict(S.synth.fill_params([(k, tuple(v.shape)) for k, v in net.net_R.state_dict().items()], seed=42))
            net.to(DEV).train()
            img_full, img_aux = S.synth.phantom_pair(n, c, h, w, seed=40)
            for _ in range(3):
                net.set_input(g(img_full), g(img_aux))
                net.update()
            torch.cuda.synchronize()
            return [p.detach().cpu().clone() for m in (net.net_R, net.net_T) for p in m.parameters()]
        finally:
            S.ops.WGRAD_OVERLAP[0] = True

    a, b, serial = run(True), run(True), run(False)
    assert all(torch.equal(x, y) for x, y in zip(a, b)), "two identical runs differ"
    assert all(torch.equal(x, y) for x, y in zip(a, serial)), "side-stream weight gradients change the result"


# ------------------------------------------------------------------ hipGraph capture of the training step
def test_captured_update_matches_eager(S):
    """[round 2] CSModel.capture_update(): three replays of the captured 'Rec' step (two streams forked / joined inside the graph,
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


# ------------------------------------------------------------------------------------------- arenas / pools
def test_two_models_interleaved_do_not_share_tapes(S):
    """[round 3] A.forward, B.forward, A.backward (a validation copy or an EMA next to the trained model): every
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


def test_amax_pool_resets_on_an_unindexed_device(S):
    """[round 3] ADVICE r2 (medium): net.to(torch.device('cuda')) -- what the reference's train.py / eval.py do -- must still zero
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


# ------------------------------------------------------------------------------------------- recorded step (host-light replay)
def test_recorded_step_replays_bit_identically(S):
    """[round 3] CSModel.record_update(): one recorded 'Rec' step replayed three times (a flat loop over the recorded C-ABI calls,
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
    """[round 3] CSModel.record_forward(): the inference pass as a recorded step; replays on new data through the static inputs give the
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


# ------------------------------------------------------------------------------------------- the reference's training loop
@pytest.mark.parametrize("reg", ["Rec", "None"])
def test_reference_train_loop_on_update_auto_records_bit_identically(S, reg):
    """[round 4] /root/reference/train.py:212-217 verbatim: ``net.set_input(*batch); net.update()`` with a NEW batch every iteration.
    ``update()`` runs two steps eagerly, records the third (the recording does not advance the model) and replays from then
    on; a validation pass (``eval(); set_input; test(); train()``) in between, a learning-rate change and a change of the loss
    weight (which drops the recording) are followed.  Parameters, BatchNorm buffers, the reconstruction and the scalar losses
    are BIT-identical to the same loop with auto-recording switched off."""
    n, c, h, w = 2, 3, 48, 80
    batches = [tuple(g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=100 + i)) for i in range(8)]

    def loop(auto):
        net = _model_r4(S, w, c, reg=reg)
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
    """[round 4] A recorded C-ABI call that fails inside a replay raises (RecordedStep.replay dropped every return code)."""
    n, c, h, w = 1, 1, 32, 32
    net = _model_r4(S, w, c, chans=4).train()
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
    """[round 4] ADVICE r3 (medium): a recorded forward pass re-packs its weight images when the weights changed since its last replay
    (optimiser step, load_state_dict), and an eager forward after training replays sees the new weights."""
    n, c, h, w = 2, 1, 64, 64
    net = _model_r4(S, w, c).eval()
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
    ref = _model_r4(S, w, c).eval()
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


def test_recording_keeps_every_packed_image_it_rewrites(S):
    """[round 4] A recorded step's weight-packing launch re-packs EVERY job of the table it was recorded with -- other live models' too.  When
    such a model is freed later and its jobs are pruned, the packed buffers must stay allocated for as long as the recording lives:
    otherwise a replay writes packed weights into memory the allocator has handed to somebody else."""
    import gc
    ops = S.ops
    n, c, h, w = 1, 1, 32, 32
    other = _model_r4(S, w, c, chans=4).eval()
    xf, xa = (g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=5))
    with torch.no_grad():
        other.set_input(xf, xa)
        other.loss_all = 0
        other.forwardT()
        other.forwardR()                                    # `other`'s weights are registered with the pack registries
    net = _model_r4(S, w, c, chans=4).eval()
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
    """[round 4] Guard bands around every arena buffer (ops.ARENA_GUARD): a full training step -- odd plane sizes at the lower levels, the
    W % 4 != 0 forms, split-K scratch, rotating dy copies -- leaves all of them intact; a deliberate store past a buffer's end is
    reported.  (A store outside a buffer is harmless while the step's streams run one after the other and corrupts a neighbour
    once they overlap: the check the stream-overlap work of round 4 needed.)"""
    ops = S.ops
    before = len(ops._GUARDS)
    ops.ARENA_GUARD[0] = True
    try:
        net = _model_r4(S, w, c, chans=4).train()
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


# ---------------------------------------------------------------------- overlap on / off at the sizes the bench and config 4 run
@pytest.mark.parametrize("tag,n,c,h,w,sparsity,steps", [("bench_n8_320", 8, 1, 320, 320, 0.25, 50),
                                                        ("config4_15x640x368", 1, 15, 640, 368, 0.125, 10)])
def test_overlapped_steps_equal_serial_steps_at_full_size(S, tag, n, c, h, w, sparsity, steps):
    """[round 5] VERDICT r4 item 2(iii): the default step runs three streams (weight gradients on the side stream, the sensitivity network
    beside the alignment network); with every overlap switched off the same kernels run one after the other.  ``steps``
    optimisation steps of the 12-cascade model (new data every step; eager, eager, then replays of the auto-recorded step) must
    leave BIT-identical parameters, BatchNorm buffers, reconstructions and losses in both forms.  The library carries no
    packed-fp32 instruction any more (tests/test_abi.py), which is what made two co-resident kernels disagree in round 4."""
    batches = [tuple(g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=700 + i)) for i in range(4)]

    def loop(overlap):
        S.ops.WGRAD_OVERLAP[0] = overlap
        S.model.SENS_OVERLAP[0] = overlap
        try:
            net = _model_r5(S, w, c, sparsity=sparsity, num_cascades=12).train()
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
    """[round 5] ``update()`` tries to record the third step.  When the recording raises (here: after the warm-up step and the recorded step
    have both run and changed the weights), the model must be put back before the call falls back to the eager step -- otherwise
    the same batch gets three optimiser steps.  Four steps with a failing recorder == four eager steps, bit for bit."""
    n, c, h, w = 2, 3, 48, 80
    batches = [tuple(g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=300 + i)) for i in range(4)]

    def loop(break_recorder):
        net = _model_r5(S, w, c, num_cascades=2, chans=6, sens_chans=4, pools=2, sens_pools=2).train()
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
    """[round 5] Queued weight gradients snapshot their operands (ops._on_side_stream): batches of 1, 4 and 16 launches behind one event give
    the same bits."""
    n, c, h, w = 2, 3, 48, 80
    batch = tuple(g(t) for t in S.synth.phantom_pair(n, c, h, w, seed=77))

    def run(k):
        old = S.ops.WGRAD_BATCH[0]
        S.ops.WGRAD_BATCH[0] = k
        try:
            net = _model_r5(S, w, c, num_cascades=2, chans=6, sens_chans=4, pools=2, sens_pools=2).train()
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
