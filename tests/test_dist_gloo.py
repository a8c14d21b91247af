"""world_size-2 gloo test (CPU) of the slice-sharding helpers the multi-GPU bench uses."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from spatialalignmentnetwork_amd import dist as sdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = sdist.init("gloo")
    assert d is not None and d.get_world_size() == world
    lo, hi = sdist.shard_bounds(total, rank, world)
    # every rank "processes" its shard: value of slice i is i; time of rank r is 1 + r
    local_sum = float(sum(range(lo, hi)))
    mean = sdist.gather_metric_mean(local_sum, hi - lo, d)
    tmax = sdist.max_over_ranks(1.0 + rank, d)
    tot = sdist.sum_over_ranks(float(hi - lo), d)
    d.barrier()
    out[rank] = (lo, hi, mean, tmax, tot)
    d.destroy_process_group()


@pytest.mark.parametrize("total", [16, 17, 3])
def test_shards_and_reductions_world2(total):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, total, out), nprocs=world, join=True)
    owned = []
    for r in range(world):
        lo, hi, mean, tmax, tot = out[r]
        owned += list(range(lo, hi))
        assert abs(mean - (total - 1) / 2.0) < 1e-12          # global mean of 0..total-1
        assert tmax == 2.0                                     # slowest rank
        assert tot == total
    assert sorted(owned) == list(range(total))                 # each slice exactly once


def test_shard_bounds_properties():
    for total in range(0, 40):
        for world in (1, 2, 3, 8):
            spans = [sdist.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = sdist.init("gloo")
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    bucket = sdist.GradBucket(lin.parameters())
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in lin.parameters())
    bucket.zero()
    for i, p in enumerate(lin.parameters()):
        p.grad.add_(float(rank + 1) * (i + 1))          # rank-dependent "gradient", written through the views
    bucket.allreduce_mean(d)
    out[rank] = [float(p.grad.flatten()[0]) for p in lin.parameters()]
    d.destroy_process_group()


def test_grad_bucket_allreduce_mean_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_grad_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert out[r] == [1.5 * (i + 1) for i in range(4)]   # mean of (1, 2) * (i + 1) on every rank


def test_param_bucket_views_cpu():
    """ParamBucket: parameters and gradients become views of flat buffers (16-byte aligned starts), values and
    Parameter identity preserved; owns() notices re-allocated parameters."""
    import torch
    from spatialalignmentnetwork_amd.dist import ParamBucket
    lin = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Conv2d(5, 2, 1))
    before = {k: v.clone() for k, v in lin.state_dict().items()}
    params = list(lin.parameters())
    b = ParamBucket(params)
    assert all(o % 4 == 0 for o in b.offsets) and b.total >= sum(p.numel() for p in params)
    for k, v in lin.state_dict().items():
        assert torch.equal(v, before[k])
    for p, o in zip(params, b.offsets):
        assert p.data_ptr() == b.flat_p.data_ptr() + 4 * o and p.grad.data_ptr() == b.flat.data_ptr() + 4 * o
    b.flat_p.mul_(2.0)                                   # one flat op updates every parameter
    assert torch.equal(lin[0].weight.detach(), 2 * before["0.weight"])
    assert b.owns(lin.parameters())
    lin.double().float()                                 # re-allocates the parameter storage
    assert not b.owns(lin.parameters())


def _step_worker(rank, world, port, out):
    """The data-parallel part of CSModel.update() on CPU: ParamBucket gradients summed over ranks in place, the
    1/world factor applied by the optimiser (here: a plain SGD stand-in for the fused kernel's grad_scale)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = sdist.init("gloo")
    torch.manual_seed(0)                                   # identical replicas
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 3, 3), torch.nn.Conv2d(3, 1, 1))
    bucket = sdist.ParamBucket(net.parameters())
    bucket.zero()
    for i, p in enumerate(net.parameters()):
        p.grad.add_(float(rank + 1) * (i + 1))
    bucket.allreduce_sum(d)
    scale = 1.0 / d.get_world_size()
    bucket.flat_p.add_(bucket.flat, alpha=-0.1 * scale)    # p -= lr * mean gradient, one flat op
    out[rank] = (bucket.flat.clone(), bucket.flat_p.clone())
    d.destroy_process_group()


def test_param_bucket_data_parallel_step_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_step_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    g0, p0 = out[0]
    g1, p1 = out[1]
    assert torch.equal(g0, g1) and torch.equal(p0, p1)     # replicas stay bit-identical after the exchange
    assert float(g0[0]) == 3.0                              # sum over ranks of (rank + 1) * 1


def test_bench_self_launches_under_torch_distributed_run():
    """`python bench.py --gpus 2` outside torch.distributed.run re-launches itself with one process per GPU
    (127.0.0.1 rendezvous) and rank 0 prints ONE JSON line; --launch-test keeps it on the CPU (gloo)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-test", "--batch", "8"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    ncores = len(os.sched_getaffinity(0))
    assert out == {"launch_test": True, "n_gpus": 2, "backend": "gloo", "max_over_ranks": 2.0, "global_batch": 16.0,
                   "pinned_cores_total": 2 * (ncores // 2)}          # every rank pinned to its own share of the cores
    # the driver's own form (torch.distributed.run around bench.py) gives the same line
    port = _free_port()
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-test",
                         "--batch", "8"], env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert [json.loads(ln) for ln in r2.stdout.splitlines() if ln.startswith("{")] == [out]


def test_bench_launches_eight_ranks():
    """The driver's 8-GPU form of the launcher on the CPU (gloo): eight ranks rendezvous, every rank pins itself to a
    disjoint core set, rank 0 prints one line for the global batch of 64 (BASELINE configs[2])."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-test", "--batch", "8"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    ncores = len(os.sched_getaffinity(0))
    assert lines == [{"launch_test": True, "n_gpus": 8, "backend": "gloo", "max_over_ranks": 8.0, "global_batch": 64.0,
                      "pinned_cores_total": 8 * (ncores // 8)}]


def test_pin_cores_gives_disjoint_shares():
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    before = os.sched_getaffinity(0)
    try:
        world = min(4, len(before))
        shares = []
        for r in range(world):
            os.sched_setaffinity(0, before)
            shares.append(set(bench.pin_cores(r, world) or []))
            assert os.sched_getaffinity(0) == shares[-1] or world == 1
        if world > 1:
            assert all(shares) and sum(len(x) for x in shares) == len(set().union(*shares))
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(max(1, min(len(before), 8)))


def test_bench_refuses_a_world_size_mismatch():
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-test"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def _bcast_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = sdist.init("gloo")
    torch.manual_seed(rank)
    t = torch.randn(5, 3)
    b = torch.tensor([rank == 0, True, rank == 1])
    nc = torch.randn(4, 6).t()                      # non-contiguous view
    sdist.broadcast0(t, d)
    sdist.broadcast0(b, d)
    sdist.broadcast0(nc, d)
    out[rank] = (t.clone(), b.clone(), nc.clone())
    d.destroy_process_group()


def test_broadcast0_world2():
    """sync_replicas' primitive: float, bool and non-contiguous tensors end up equal to rank 0's on every rank."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bcast_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    torch.manual_seed(0)
    want_t = torch.randn(5, 3)
    want_nc = torch.randn(4, 6).t()
    for r in range(2):
        t, b, nc = out[r]
        assert torch.equal(t, want_t) and torch.equal(b, torch.tensor([True, True, False])) and torch.equal(nc, want_nc)


def _cascade_bucket_worker(rank, world, port, out):
    """Per-cascade reverse-order gradient exchange against the single-buffer form, on the real VarNet
    parameter layout: the slices partition the flat buffer, go out in the order (cascade T-1 .. 0, sens) and leave the
    same bits behind as one all-reduce of the whole buffer."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = sdist.init("gloo")
    import types
    from spatialalignmentnetwork_amd.model import CSModel
    from spatialalignmentnetwork_amd.varnet import VarNet
    torch.manual_seed(0)
    net_R = VarNet(num_cascades=3, sens_chans=2, sens_pools=1, chans=4, pools=1, use_ref=True)
    bucket = sdist.ParamBucket(net_R.parameters())
    ranges = CSModel._cascade_ranges(types.SimpleNamespace(net_R=net_R), bucket)
    g = torch.Generator().manual_seed(100 + rank)
    grads = torch.randn(bucket.total, generator=g)
    # (a) per cascade, in the order VarNet.backward visits them
    bucket.flat.copy_(grads)
    exch = sdist.GradExchange(d)
    for which in [2, 1, 0, "sens"]:
        exch.launch(bucket, rng=ranges[which])
    exch.wait()
    per_cascade = bucket.flat.clone()
    # (b) the whole buffer at once
    bucket.flat.copy_(grads)
    exch1 = sdist.GradExchange(d)
    exch1.launch(bucket)
    exch1.wait()
    out[rank] = (per_cascade, bucket.flat.clone(), list(exch.launched), ranges, bucket.total)
    d.destroy_process_group()


def test_per_cascade_buckets_match_the_single_buffer_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_cascade_bucket_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        per_cascade, single, launched, ranges, total = out[r]
        assert torch.equal(per_cascade, single)
        assert launched == [ranges[2], ranges[1], ranges[0], ranges["sens"]]
        spans = sorted(launched)
        assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        # the sensitivity net comes first in parameter order, the cascades follow in order
        assert ranges["sens"][0] == 0 and ranges[0][0] == ranges["sens"][1] and ranges[2][1] == total
    assert torch.equal(out[0][0], out[1][0])


# ------------------------------------------------------------------ round 6: more than two ranks (gloo, CPU)
def _many_ranks_worker(rank, world, port, out):
    """What a data-parallel job does around its first step, on `world` CPU ranks: every rank builds its OWN randomly initialised
    CSModel (different seeds), sync_replicas() makes them rank 0's, the per-cascade gradient exchange sums rank-dependent
    gradients slice by slice in the order VarNet.backward releases them, and the ranks agree on whether a recording failed."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = sdist.init("gloo")
    from spatialalignmentnetwork_amd import basemodel, model
    torch.manual_seed(1000 + rank)
    cfg = basemodel.Config(sparsity=0.25, lr=1e-4, shape=32, coils=1, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                           weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=3, chans=4,
                           sens_chans=2, pools=1, sens_pools=1)
    net = model.CSModel(cfg)
    bR, bT = net.optim_R.bucket(), net.optim_T.bucket()
    bR.exp_avg.fill_(float(rank))                       # (moments, step counts and BatchNorm buffers travel too)
    bR.steps = 7 * rank
    for buf in net.net_T.buffers():
        if buf.dtype.is_floating_point:
            buf.add_(rank)
    before = bR.flat_p.clone()
    net.sync_replicas(d)
    state = torch.cat([bR.flat_p, bT.flat_p, bR.exp_avg, net.net_mask.weight.data.float().reshape(-1),
                       net.net_mask.pruned.float().reshape(-1)] + [b.float().reshape(-1) for b in net.net_T.buffers()])
    # the exchange: rank-dependent gradients, per cascade in reverse order, then the sensitivity net's slice, then net_T's buffer
    ranges = net._cascade_ranges(bR)
    gen = torch.Generator().manual_seed(500 + rank)
    bR.flat.copy_(torch.randn(bR.total, generator=gen))
    bT.flat.copy_(torch.randn(bT.total, generator=gen))
    exch = sdist.GradExchange(d)
    for which in [2, 1, 0, "sens"]:
        exch.launch(bR, rng=ranges[which])
    exch.launch(bT)
    exch.wait()
    # agreement on a failed recording: nobody failed -> None everywhere; the LAST rank failed -> a message everywhere
    ok_all = sdist.agree_on_failure(None, d, "cpu")
    one_bad = sdist.agree_on_failure("boom" if rank == world - 1 else None, d, "cpu")
    out[rank] = (state, bR.steps, bR.flat.clone(), bT.flat.clone(), list(exch.launched), ranges, ok_all, one_bad,
                 bool(torch.equal(before, bR.flat_p)), sdist.gather_over_ranks(10.0 + rank, d))
    sdist.shutdown()
    d.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_replica_sync_per_cascade_exchange_and_agreement_on_many_ranks(world):
    """gloo world-4 and world-8 (world 2 above): after sync_replicas every rank holds rank 0's parameters,
    moments, step count, mask and BatchNorm buffers; the per-cascade slices partition net_R's flat buffer and every rank ends
    with the same bits, equal to the float64 sum of all ranks' gradients to fp32 rounding; agree_on_failure gives every rank
    the same verdict; gather_over_ranks returns the per-rank values in rank order."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_many_ranks_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    state0, steps0 = out[0][0], out[0][1]
    assert steps0 == 0 and out[0][8]                     # rank 0 keeps its own state
    wantR = wantT = None
    for r in range(world):
        gen = torch.Generator().manual_seed(500 + r)
        gR = torch.randn(out[0][2].numel(), generator=gen).double()
        gT = torch.randn(out[0][3].numel(), generator=gen).double()
        wantR = gR if wantR is None else wantR + gR
        wantT = gT if wantT is None else wantT + gT
    for r in range(world):
        state, steps, fR, fT, launched, ranges, ok_all, one_bad, unchanged, gathered = out[r]
        assert torch.equal(state, state0) and steps == steps0
        assert r == 0 or not unchanged                   # (the other ranks really started from different weights)
        assert torch.equal(fR, out[0][2]) and torch.equal(fT, out[0][3])       # every rank ends with the same bits
        assert (fR.double() - wantR).abs().max().item() < 1e-5 and (fT.double() - wantT).abs().max().item() < 1e-5
        assert launched == [ranges[2], ranges[1], ranges[0], ranges["sens"], None]
        spans = sorted(x for x in launched if x is not None)
        assert spans[0][0] == 0 and spans[-1][1] == fR.numel() and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert ok_all is None and isinstance(one_bad, str)
        assert one_bad == ("boom" if r == world - 1 else "another rank could not record the step")
        assert gathered == [10.0 + k for k in range(world)]


def test_exchange_mode_switch_is_validated(monkeypatch):
    monkeypatch.setenv("SAN_GRAD_EXCHANGE", "rs_ag")
    assert sdist.exchange_mode() == "rs_ag"
    monkeypatch.setenv("SAN_GRAD_EXCHANGE", "ring")
    with pytest.raises(ValueError):
        sdist.exchange_mode()
    monkeypatch.delenv("SAN_GRAD_EXCHANGE")
    assert sdist.exchange_mode() == "allreduce"
