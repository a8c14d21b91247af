"""GPU tests (through the C ABI) of the normalisation FORWARD family: the statistics a convolution emits and their finalisation
into the lazy affine (san_norm_finalize), alone and fused with the encoder levels' average pooling (san_norm_finalize_pool, round 6).
Reference arithmetic: conv -> InstanceNorm2d -> LeakyReLU -> avg_pool2d (varnet.py:95-99, 139-146).
(Round 6 also built the finalisation INSIDE the producing convolution by the last workgroup of a reduction domain; it was 4-34 us
slower per layer than the separate launch and is not in the library: scratch/attempts/r6_fin_inkernel.patch, r6_notes.md.)"""
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
    from spatialalignmentnetwork_amd import ops

    class NS:
        pass

    ns = NS()
    ns.ops = ops
    return ns


def g(t):
    return t.to(DEV).contiguous()


@pytest.mark.parametrize("n,c,h,w", [(8, 18, 320, 320), (8, 36, 160, 160), (4, 72, 80, 80), (2, 144, 40, 24), (3, 5, 16, 32)])
def test_finalisation_and_average_pooling_in_one_launch_is_bit_identical(S, n, c, h, w):
    """san_norm_finalize_pool (round 6) == san_norm_finalize + san_avgpool2_fwd bit for bit: the affine of the convolution's records
    and avg_pool2d(lrelu(IN(y))) (the U-Net encoder levels, varnet.py:95-99), also through channel views."""
    ops = S.ops
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
