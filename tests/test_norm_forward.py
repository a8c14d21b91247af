"""GPU tests (through the C ABI) of the normalisation FORWARD family: the statistics a convolution emits and their finalisation
into the lazy affine -- in a launch of its own (san_norm_finalize / san_norm_finalize_bn) and, round 6, INSIDE the producing
convolution by the last workgroup of a reduction domain (san_conv_bf16x3_fwd_fin / _fin_bn; csrc/san_fin.h: correct, but
slower than the second launch on this hardware and therefore off by default -- the tests switch it on explicitly).
Reference arithmetic: conv -> InstanceNorm2d (varnet.py:139-146, 171-176), conv -> training BatchNorm2d (unet.py:119-126)."""
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
    default = ops.FIN_INKERNEL[0]
    yield ns
    ops.FIN_INKERNEL[0] = default


def g(t):
    return t.to(DEV).contiguous()


def _in_affine64(y):
    mean = y.double().mean((2, 3))
    var = y.double().var((2, 3), unbiased=False)
    sc = 1.0 / torch.sqrt(var + 1e-5)
    return sc, -mean * sc


# (n, cin, cout, h, w, ks): the persistent stream kernel (320^2 / 160^2, 18 / 36 channels), the tiled kernel (80^2, 40^2 full-width
# tiles), a split-K layer (20^2), 1x1 GEMM layers, odd sizes with partial tiles, one sample spread over all XCDs
CONV_SHAPES = [(8, 18, 18, 320, 320, 3), (8, 36, 18, 320, 320, 3), (8, 36, 36, 160, 160, 3), (3, 18, 36, 160, 160, 3),
               (1, 18, 18, 320, 320, 3), (8, 72, 72, 80, 80, 3), (8, 144, 144, 40, 40, 3), (8, 288, 288, 20, 20, 3),
               (2, 64, 64, 48, 80, 3), (2, 36, 72, 72, 56, 3), (4, 64, 32, 160, 160, 1), (2, 96, 64, 40, 24, 1)]


@pytest.mark.parametrize("n,cin,cout,h,w,ks", CONV_SHAPES)
def test_instance_norm_finalised_inside_the_convolution(S, n, cin, cout, h, w, ks):
    """conv2d(..., instance_norm_eps): the affine written by the last workgroup of each sample equals float64 InstanceNorm2d
    statistics of the convolution's own output and the separate san_norm_finalize launch; the launch reports that it finalised
    (returns no records) for every layer kind; the output tensor is the same bits either way."""
    ops = S.ops
    x = g(philox("nf.x", (n, cin, h, w)) * 2)
    wt = g(philox("nf.w", (cout, cin, ks, ks)) * (0.5 / (cin * ks * ks) ** 0.5) * 4)
    res = {}
    for on in (False, True):
        ops.FIN_INKERNEL[0] = on
        y = ops.Act(torch.full((n, cout, h, w), float("nan"), device=DEV), 0, cout, torch.full((n, cout), float("nan"), device=DEV),
                    torch.full((n, cout), float("nan"), device=DEV), 0.2)
        part = ops.conv2d(ops.full(x), wt, None, y, stats=True, instance_norm_eps=1e-5, tag=".nf")
        if on:
            assert part is None, "the launch should have finalised the affine itself"
        if part is not None:
            ops.norm_finalize(part, ops.NORM_INSTANCE, 1e-5, y.scale, y.shift, y.coff)
        torch.cuda.synchronize()
        res[on] = y
    assert torch.equal(res[False].buf, res[True].buf)
    sc64, sh64 = _in_affine64(res[True].buf)
    for on in (False, True):
        assert ((res[on].scale.double() - sc64).abs() / sc64.abs()).max().item() < 2e-6
        assert ((res[on].shift.double() - sh64).abs().max() / sh64.abs().max()).item() < 2e-6
    assert ((res[True].scale - res[False].scale).abs() / res[False].scale.abs()).max().item() < 5e-7


@pytest.mark.parametrize("n,cin,cout,h,w", [(8, 36, 18, 160, 160), (8, 288, 144, 20, 20), (2, 72, 36, 40, 24)])
def test_transposed_convolution_finalises_its_instance_norm(S, n, cin, cout, h, w):
    """tconv2x2(..., instance_norm_eps): a real channel's statistic is the union of its four virtual channels' records."""
    ops = S.ops
    x = g(philox("nt.x", (n, cin, h, w)) * 2)
    wt = g(philox("nt.w", (cin, cout, 2, 2)) * 0.2)
    want = F.conv_transpose2d(x.double(), wt.double(), stride=2)
    sc64, sh64 = _in_affine64(want)
    for on in (False, True):
        ops.FIN_INKERNEL[0] = on
        y = ops.Act(torch.empty((n, cout, 2 * h, 2 * w), device=DEV), 0, cout, torch.full((n, cout), float("nan"), device=DEV),
                    torch.full((n, cout), float("nan"), device=DEV), 0.2)
        part = ops.tconv2x2(ops.full(x), wt, y, stats=True, tag=".nt", instance_norm_eps=1e-5)
        assert (part is None) == on
        if part is not None:
            ops.norm_finalize(part, ops.NORM_INSTANCE, 1e-5, y.scale, y.shift, y.coff)
        torch.cuda.synchronize()
        assert ((y.buf.double() - want).abs().max() / want.abs().max()).item() < 3e-6
        assert ((y.scale.double() - sc64).abs() / sc64.abs()).max().item() < 5e-6
        assert ((y.shift.double() - sh64).abs().max() / sh64.abs().max()).item() < 5e-6


@pytest.mark.parametrize("n,cin,cout,h,w,ks", [(8, 32, 32, 320, 320, 3), (8, 64, 64, 160, 160, 3), (4, 64, 64, 40, 40, 3),
                                              (8, 96, 32, 320, 320, 3), (8, 64, 32, 160, 160, 1), (3, 64, 64, 20, 20, 3)])
def test_batch_norm_finalised_inside_the_convolution(S, n, cin, cout, h, w, ks):
    """conv2d(..., batch_norm=...): ONE reduction domain over the batch -- affine, batch mean / unbiased variance, running
    statistics and num_batches_tracked equal san_norm_finalize_bn's and torch's training BatchNorm2d in float64."""
    ops = S.ops
    x = g(philox("nb.x", (n, cin, h, w)) * 2)
    wt = g(philox("nb.w", (cout, cin, ks, ks)) * (2.0 / (cin * ks * ks) ** 0.5))
    bias = g(philox("nb.b", (cout,)) * 0.3)
    out = {}
    for on in (False, True):
        ops.FIN_INKERNEL[0] = on
        bn = torch.nn.BatchNorm2d(cout).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(g(philox("nb.ga", (cout,), lo=0.5, hi=1.5)))
            bn.bias.copy_(g(philox("nb.be", (cout,)) * 0.2))
            bn.running_mean.copy_(g(philox("nb.rm", (cout,)) * 0.1))
            bn.running_var.copy_(g(philox("nb.rv", (cout,), lo=0.5, hi=1.5)))
        y = ops.Act(torch.empty((n, cout, h, w), device=DEV), 0, cout, torch.full((n, cout), float("nan"), device=DEV),
                    torch.full((n, cout), float("nan"), device=DEV), 0.01)
        bmean, bvar = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
        part = ops.conv2d(ops.full(x), wt, bias, y, stats=True, tag=".nb", batch_norm=(1e-5, bn, bmean, bvar, 0.1, 1.0))
        assert (part is None) == on
        if part is not None:
            ops.norm_finalize_bn(part, 1e-5, y.scale, y.shift, y.coff, bn, bmean, bvar, 0.1, 1.0)
        torch.cuda.synchronize()
        out[on] = (y, bmean, bvar, bn)
    y, bmean, bvar, bn = out[True]
    ref = torch.nn.BatchNorm2d(cout).to(DEV).double()
    with torch.no_grad():
        ref.weight.copy_(bn.weight)
        ref.bias.copy_(bn.bias)
        ref.running_mean.copy_(g(philox("nb.rm", (cout,)) * 0.1))
        ref.running_var.copy_(g(philox("nb.rv", (cout,), lo=0.5, hi=1.5)))
    ref.train()
    want = ref(y.buf.double())
    got = y.buf.double() * y.scale.double()[:, :, None, None] + y.shift.double()[:, :, None, None]
    assert ((got - want).abs().max() / want.abs().max()).item() < 3e-6
    assert ((bn.running_mean.double() - ref.running_mean).abs().max()).item() < 1e-6
    assert ((bn.running_var.double() - ref.running_var).abs() / ref.running_var).max().item() < 2e-6
    assert int(bn.num_batches_tracked) == 1
    y0, bmean0, bvar0, bn0 = out[False]
    for a, b in ((y.scale, y0.scale), (y.shift, y0.shift), (bmean, bmean0), (bvar, bvar0), (bn.running_var, bn0.running_var)):
        assert ((a - b).abs().max() / b.abs().max()).item() < 1e-6


def test_tickets_return_to_zero_and_every_launch_sees_its_own_records(S):
    """Alternating inputs through the same statistics buffers and ticket words, 120 launches of a stream-kernel layer and a tiled
    layer: every launch's affine belongs to ITS input (nothing stale), repeats bit for bit, and the ticket words are zero again."""
    ops = S.ops
    ops.FIN_INKERNEL[0] = True
    cases = []
    for (n, c, h, w) in [(8, 18, 320, 320), (8, 72, 80, 80)]:
        wt = g(philox("tk.w", (c, c, 3, 3)) * 0.1)
        xs = [g(philox(f"tk.x{k}", (n, c, h, w)) * (1.0 + 2.0 * k) + 0.5 * k) for k in range(2)]
        cases.append((n, c, h, w, wt, xs))
    first = {}
    for it in range(120):
        k = it & 1
        for ci, (n, c, h, w, wt, xs) in enumerate(cases):
            y = ops.Act(torch.empty((n, c, h, w), device=DEV), 0, c, torch.empty((n, c), device=DEV), torch.empty((n, c), device=DEV), 0.2)
            assert ops.conv2d(ops.full(xs[k]), wt, None, y, stats=True, instance_norm_eps=1e-5, tag=".tk") is None
            if it < 2 or it % 20 < 2:
                torch.cuda.synchronize()
                sc64, sh64 = _in_affine64(y.buf)
                assert ((y.scale.double() - sc64).abs() / sc64.abs()).max().item() < 2e-6
                key = (ci, k)
                if key not in first:
                    first[key] = (y.scale.clone(), y.shift.clone())
                assert torch.equal(first[key][0], y.scale) and torch.equal(first[key][1], y.shift)
    torch.cuda.synchronize()
    tk = ops.GLOBAL_ARENA.get("fin_ticket", (8,), torch.device(DEV), dtype=torch.int32, zero=True)
    assert int(tk.abs().sum()) == 0


@pytest.mark.parametrize("n,c,h,w", [(8, 18, 320, 320), (8, 36, 160, 160), (4, 72, 80, 80), (2, 144, 40, 24), (3, 5, 16, 32)])
def test_finalisation_and_average_pooling_in_one_launch_is_bit_identical(S, n, c, h, w):
    """san_norm_finalize_pool (round 6) == san_norm_finalize + san_avgpool2_fwd bit for bit: the affine of the convolution's records
    and avg_pool2d(lrelu(IN(y))) (the U-Net encoder levels, varnet.py:95-99), also through channel views."""
    ops = S.ops
    ops.FIN_INKERNEL[0] = False
    x = g(philox("fp.x", (n, c, h, w)) * 2)
    wt = g(philox("fp.w", (c, c, 3, 3)) * 0.1)
    outs = []
    for fused in (False, True):
        cat = ops.Act(torch.zeros((n, 2 * c, h, w), device=DEV), 0, 2 * c, torch.zeros((n, 2 * c), device=DEV), torch.zeros((n, 2 * c), device=DEV), 0.2)
        y = cat.view(c, c)
        pooled = ops.Act(torch.full((n, c + 1, h // 2, w // 2), -7.0, device=DEV), 1, c)
        part = ops.conv2d(ops.full(x), wt, None, y, stats=True, tag=".fp")
        if part is None:
            pytest.skip("this layer finalises in its split-K reduction")
        if fused:
            ops.norm_finalize_pool(part, 1e-5, y, pooled)
        else:
            ops.norm_finalize(part, ops.NORM_INSTANCE, 1e-5, y.scale, y.shift, y.coff)
            ops.avgpool2(y, pooled)
        torch.cuda.synchronize()
        outs.append((cat.scale.clone(), cat.shift.clone(), pooled.buf.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(outs[1][2][:, 0].min()) == -7.0 == float(outs[1][2][:, 0].max())           # the channel outside the view is untouched
    want = F.avg_pool2d(F.leaky_relu(F.instance_norm(F.conv2d(x.double(), wt.double(), padding=1), eps=1e-5), 0.2), 2)
    assert ((outs[1][2][:, 1:].double() - want).abs().max() / want.abs().max()).item() < 5e-6
