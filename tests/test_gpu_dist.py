"""GPU parity tests (through the C ABI), component: multi-rank paths that need the GPU (gloo on one GPU, one-rank RCCL) (row e).
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


def test_update_data_parallel_two_ranks(S, tmp_path):
    """[round 2] CSModel.update()'s data-parallel branch (flat-buffer all-reduce, 1/world inside AdamW, replica sync from rank 0)
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


def test_captured_step_under_a_process_group_two_ranks(S, tmp_path):
    """[round 3] VERDICT r2 #4a: capture_update works with an active process group.  Two ranks (gloo, one GPU): the captured step
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


def test_gradient_exchange_runs_on_rccl_with_one_rank(S, tmp_path):
    """[round 4] VERDICT r3: 'RCCL code paths have literally never run.'  A one-GPU box cannot show the transport, but it can run every
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


# --------------------------------------------------------------------------------- bench.py through the launcher on one-rank RCCL
def test_bench_launcher_runs_the_exchange_on_rccl_with_one_rank():
    """[round 5] VERDICT r4 item 7: ``python bench.py --gpus 1`` with SAN_DIST_SINGLE=1 brings up a one-rank RCCL process group and runs the
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
