"""GPU tests (through the C ABI) of the normalisation + activation BACKWARD family: the one-pass forms on workgroup
clusters (round 6: san_act_bwd_in, san_bn_act_bwd) against float64 / torch autograd and against the multi-launch forms they
replace.  Reference arithmetic: InstanceNorm2d + LeakyReLU (varnet.py:139-146), training BatchNorm2d + LeakyReLU (unet.py:125)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import philox

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def S():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spatialalignmentnetwork_amd import ops, _lib

    class NS:
        pass

    ns = NS()
    ns.ops, ns.lib = ops, _lib.lib()
    yield ns
    ns.lib.call("san_act_bwd_cluster_set_tuning", 1, 160 * 160, 4, 1)         # (the shipped defaults)


def g(t):
    return t.to(DEV).contiguous()


def _tune(S, on=-1, min_hw=0, v=0, bn=-1):
    S.lib.call("san_act_bwd_cluster_set_tuning", on, min_hw, v, bn)


def _in_autograd(gin, y, sc, sh, slope, g2=None):
    """dL/dy_raw of a = lrelu(IN-affine(y)) by torch autograd in float64, with the affine (sc, sh) DERIVED from y's plane
    statistics (InstanceNorm2d, biased variance), i.e. the real dependency the hand-written backward differentiates."""
    y64 = y.double().requires_grad_(True)
    mean = y64.mean((2, 3), keepdim=True)
    var = y64.var((2, 3), unbiased=False, keepdim=True)
    a = F.leaky_relu((y64 - mean) / torch.sqrt(var + 1e-5), slope)
    gg = gin.double()
    if g2 is not None:
        gg = gg + 0.25 * F.interpolate(g2.double(), scale_factor=2, mode="nearest")
    (a * gg).sum().backward()
    return y64.grad


@pytest.mark.parametrize("n,c,h,w,with_g2,min_hw,v", [
    (8, 18, 320, 320, False, 25600, 4), (8, 18, 320, 320, True, 25600, 4), (2, 3, 320, 320, False, 25600, 7),
    (4, 36, 160, 160, True, 25600, 4), (1, 5, 640, 368, False, 25600, 7), (3, 7, 96, 72, True, 1025, 2),
    (2, 4, 48, 80, False, 513, 1)])
def test_instance_norm_backward_on_clusters_vs_autograd_and_two_launch_form(S, n, c, h, w, with_g2, min_hw, v):
    """san_act_bwd_in (K workgroups per plane share the two plane sums through a sync record) equals float64 autograd through
    InstanceNorm2d + LeakyReLU(0.2) and the bwd_stats + act_bwd pair, plain and with the half-resolution second gradient source."""
    ops = S.ops
    gin = g(philox("nb.g", (n, c, h, w)) * 1e-3)
    y = g(philox("nb.y", (n, c, h, w)) * 2 + 0.3)
    g2 = g(philox("nb.g2", (n, c, h // 2, w // 2)) * 1e-3) if with_g2 else None
    mean = y.double().mean((2, 3))
    var = y.double().var((2, 3), unbiased=False)
    sc = (1.0 / torch.sqrt(var + 1e-5)).float().contiguous()
    sh = (-mean / torch.sqrt(var + 1e-5)).float().contiguous()
    want = _in_autograd(gin, y, sc, sh, 0.2, g2)
    outs = []
    for on in (0, 1):
        _tune(S, on=on, min_hw=min_hw, v=v)
        assert bool(S.lib.query("san_act_bwd_in_sync_words", n, c, h * w)) == bool(on)
        ops.AMAX.reset(DEV)
        dy = ops.Act(torch.full((n, c, h, w), float("nan"), device=DEV), 0, c)
        ops.act_bwd(ops.full(gin), ops.Act(y, 0, c, sc, sh, 0.2), dy, instance_norm=True, g2=None if g2 is None else ops.full(g2))
        torch.cuda.synchronize()
        outs.append(dy)
    scale = want.abs().max().item()
    for dy in outs:
        assert (dy.buf.double() - want).abs().max().item() / scale < 5e-6
    # the amax record of the one-pass form holds the tensor's largest magnitude (the fp16-part gradient kernels scale by it)
    if outs[1].amax is not None:
        assert abs(ops.amax_value(outs[1].amax) - outs[1].buf.abs().max().item()) <= 1e-12


@pytest.mark.parametrize("mode", ["unshuffle", "accumulate", "views"])
def test_cluster_form_destinations_and_channel_views(S, mode):
    """The pixel-unshuffled and accumulated destinations (san_act_bwd_ex_amax's flags) and channel views of all three tensors:
    the cluster form equals the two-launch form."""
    ops = S.ops
    n, c, h, w = 4, 18, 320, 320
    res = []
    for on in (0, 1):
        _tune(S, on=on, min_hw=25600, v=4)
        ops.AMAX.reset(DEV)
        if mode == "views":
            gbuf, ybuf = g(philox("nb.vg", (n, c + 5, h, w))), g(philox("nb.vy", (n, 2 * c, h, w)))
            sc, sh = g(philox("nb.vs", (n, 2 * c), lo=0.5, hi=1.5)), g(philox("nb.vh", (n, 2 * c)) * 0.3)
            dbuf = torch.zeros((n, c + 3, h, w), device=DEV)
            ops.act_bwd(ops.Act(gbuf, 5, c), ops.Act(ybuf, c, c, sc, sh, 0.2), ops.Act(dbuf, 2, c), instance_norm=True)
            assert float(dbuf[:, :2].abs().max()) == 0.0 and float(dbuf[:, 2 + c:].abs().max()) == 0.0
            res.append(dbuf)
            continue
        gin, y = g(philox("nb.dg", (n, c, h, w))), g(philox("nb.dy", (n, c, h, w)))
        sc, sh = g(philox("nb.ds", (n, c), lo=0.5, hi=1.5)), g(philox("nb.dh", (n, c)) * 0.3)
        if mode == "unshuffle":
            dy = torch.zeros(n, 4 * c, h // 2, w // 2, device=DEV)
            ops.act_bwd_ex(ops.full(gin), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True, unshuffle=True)
        else:
            dy = torch.ones(n, c, h, w, device=DEV)
            ops.act_bwd_ex(ops.full(gin), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True, accumulate=True)
        res.append(dy)
    torch.cuda.synchronize()
    assert ((res[0] - res[1]).abs().max() / res[0].abs().max()).item() < 5e-6


@pytest.mark.parametrize("n,c,h,w", [(8, 64, 160, 160), (2, 32, 320, 320), (8, 64, 80, 80), (8, 64, 20, 20), (3, 5, 40, 24), (8, 32, 320, 320)])
def test_batch_norm_backward_on_clusters_vs_autograd_and_three_launch_form(S, n, c, h, w):
    """san_bn_act_bwd (one cluster per channel spanning the batch) against float64 autograd through training-mode
    BatchNorm2d + LeakyReLU(0.01) (unet.py:125) and against san_plane_dot_stats + san_bn_bwd_finalize + san_act_bwd_coef_amax;
    (8, 32, 320, 320) has more members than a cluster takes and must fall back to the three launches by itself."""
    ops = S.ops
    gin = g(philox("bn.g", (n, c, h, w)) * 1e-2)
    y = g(philox("bn.y", (n, c, h, w)) + 0.2)
    gamma, beta = g(philox("bn.ga", (c,), lo=0.5, hi=1.5)), g(philox("bn.be", (c,)) * 0.2)
    y64 = y.double().requires_grad_(True)
    ga64, be64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.leaky_relu(F.batch_norm(y64, None, None, ga64, be64, True, 0.1, 1e-5), 0.01)
    (a * gin.double()).sum().backward()
    mean, var = y.double().mean((0, 2, 3)), y.double().var((0, 2, 3), unbiased=False)
    sc1 = gamma.double() / torch.sqrt(var + 1e-5)
    sc = sc1.float()[None].repeat(n, 1).contiguous()
    sh = (beta.double() - mean * sc1).float()[None].repeat(n, 1).contiguous()
    res = []
    for bn in (0, 1):
        _tune(S, bn=bn)
        words = S.lib.query("san_bn_act_bwd_sync_words", n, c, h * w)
        assert bool(words) == (bool(bn) and (n, c, h, w) != (8, 32, 320, 320))
        ops.AMAX.reset(DEV)
        dy = torch.full((n, c, h, w), float("nan"), device=DEV)
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        ops.bn_act_bwd(ops.full(gin), ops.Act(y, 0, c, sc, sh, 0.01), gamma, beta, dg, db, ops.full(dy))
        torch.cuda.synchronize()
        res.append((dy, dg, db))
    for dy, dg, db in res:
        assert ((dy.double() - y64.grad).abs().max() / y64.grad.abs().max()).item() < 2e-5
        assert ((dg.double() - ga64.grad).abs().max() / ga64.grad.abs().max()).item() < 2e-5
        assert ((db.double() - be64.grad).abs().max() / be64.grad.abs().max()).item() < 2e-5


def test_sync_records_carry_nothing_from_one_launch_to_the_next(S):
    """Two DIFFERENT inputs alternate through the same sync buffers (and a third shape with the same word count shares the
    arena): every launch equals the multi-launch result of ITS input -- a stale partial sum or a counter left behind by another
    cluster layout would show at once; 200 launches repeat bit for bit."""
    ops = S.ops
    n, c, h, w = 2, 32, 160, 160
    sets, refs = [], []
    for k in range(2):
        sets.append((g(philox(f"st.g{k}", (n, c, h, w)) * (1.0 + 3.0 * k)), g(philox(f"st.y{k}", (n, c, h, w)) + 0.5 * k),
                     g(philox(f"st.ga{k}", (c,), lo=0.5, hi=1.5)), g(philox(f"st.be{k}", (c,)) * 0.2),
                     g(philox(f"st.sc{k}", (n, c), lo=0.5, hi=1.5)), g(philox(f"st.sh{k}", (n, c)))))

    def run(k):
        gin, y, gamma, beta, sc, sh = sets[k]
        ops.AMAX.reset(DEV)
        dy, dyi = torch.empty_like(gin), torch.empty_like(gin)
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        ops.bn_act_bwd(ops.full(gin), ops.Act(y, 0, c, sc, sh, 0.01), gamma, beta, dg, db, ops.full(dy))
        ops.act_bwd(ops.full(gin), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dyi), instance_norm=True)
        return dy, dg, db, dyi

    _tune(S, on=0, bn=0)
    for k in range(2):
        refs.append([t.clone() for t in run(k)])
    _tune(S, on=1, min_hw=1025, v=7, bn=1)
    first = [None, None]
    for it in range(200):
        k = it & 1
        if it == 100:
            _tune(S, v=2)                      # another cluster size through the same buffers mid-way: records return to all zeros
            first = [None, None]
        out = run(k)
        if it % 10 < 2:
            for a, b in zip(out, refs[k]):
                assert ((a - b).abs().max() / b.abs().max()).item() < 2e-5
            if first[k] is None:
                first[k] = [t.clone() for t in out]
            assert all(torch.equal(a, b) for a, b in zip(out, first[k]))


@pytest.mark.parametrize("n,c,h,w,with_g2", [(8, 144, 40, 40, False), (8, 144, 40, 40, True), (8, 288, 20, 20, False), (2, 7, 20, 20, True),
                                            (3, 5, 8, 16, False), (1, 3, 4, 4, False), (8, 72, 80, 80, True)])
def test_small_planes_vs_autograd(S, n, c, h, w, with_g2):
    """InstanceNorm + LeakyReLU backward of the small planes (80 x 80 down to 4 x 4: the one-workgroup plane kernel) against float64
    autograd, plain, with the half-resolution second source, and with the pixel-unshuffled store (== act_bwd + unshuffle2 bit for
    bit).  (Round 6 also tried one WAVE per plane for <= 40 x 40 -- no barrier, no LDS: correct, and no faster in the step,
    41.18 / 41.07 vs 41.03 / 41.40 ms same box; not kept.)"""
    ops = S.ops
    gin = g(philox("sw.g", (n, c, h, w)) * 1e-3)
    y = g(philox("sw.y", (n, c, h, w)) * 2 + 0.3)
    g2 = g(philox("sw.g2", (n, c, h // 2, w // 2)) * 1e-3) if with_g2 else None
    mean = y.double().mean((2, 3))
    var = y.double().var((2, 3), unbiased=False)
    sc = (1.0 / torch.sqrt(var + 1e-5)).float().contiguous()
    sh = (-mean / torch.sqrt(var + 1e-5)).float().contiguous()
    want = _in_autograd(gin, y, sc, sh, 0.2, g2)
    ops.AMAX.reset(DEV)
    dy = ops.Act(torch.full((n, c, h, w), float("nan"), device=DEV), 0, c)
    ops.act_bwd(ops.full(gin), ops.Act(y, 0, c, sc, sh, 0.2), dy, instance_norm=True, g2=None if g2 is None else ops.full(g2))
    torch.cuda.synchronize()
    assert ((dy.buf.double() - want).abs().max() / want.abs().max()).item() < 5e-6
    if dy.amax is not None:
        assert abs(ops.amax_value(dy.amax) - dy.buf.abs().max().item()) <= 1e-12
    if g2 is None and h % 2 == 0 and w % 4 == 0:
        ops.AMAX.reset(DEV)
        dyp = ops.Act(torch.zeros((n, 4 * c, h // 2, w // 2), device=DEV), 0, 4 * c)
        ops.act_bwd_ex(ops.full(gin), ops.Act(y, 0, c, sc, sh, 0.2), dyp, instance_norm=True, unshuffle=True)
        ref = ops.Act(torch.zeros((n, 4 * c, h // 2, w // 2), device=DEV), 0, 4 * c)
        ops.unshuffle2(ops.Act(dy.buf, 0, c), ref)
        torch.cuda.synchronize()
        assert torch.equal(dyp.buf, ref.buf)
