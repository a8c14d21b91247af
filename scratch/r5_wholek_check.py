"""Whole-K 3x3 convolution on narrow images (round 5) vs float64 and vs the tiled kernel (incl. its split-K form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from spatialalignmentnetwork_amd import ops
from spatialalignmentnetwork_amd.ops import Act
dev = "cuda:0"
torch.manual_seed(0)
REPS = int(os.environ.get("REPS", "20"))


def act64(x, sc, sh, slope):
    xd = x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
    return torch.where(xd >= 0, xd, xd * slope)


def stats_from(part):
    assert torch.isfinite(part).all()
    cnt, mean, m2 = part[..., 0].double(), part[..., 1].double(), part[..., 2].double()
    tot = cnt.sum(-1)
    mu = (cnt * mean).sum(-1) / tot
    var = (m2 + cnt * (mean - mu[..., None]) ** 2).sum(-1) / tot
    return mu, var, tot


worst = 0.0
for (n, cin, cout, h, w, bias_on) in [(8, 72, 144, 40, 40, False), (8, 144, 144, 40, 40, False), (8, 288, 144, 40, 40, False),
                                      (8, 144, 288, 20, 20, False), (8, 288, 288, 20, 20, True), (1, 72, 144, 80, 46, False),
                                      (1, 144, 288, 40, 23, False), (2, 40, 24, 23, 17, True), (3, 18, 36, 9, 48, False)]:
    x = torch.randn(n, cin, h, w, device=dev)
    sc, sh = torch.rand(n, cin, device=dev) + 0.5, torch.randn(n, cin, device=dev) * 0.3
    wt = torch.randn(cout, cin, 3, 3, device=dev) * (1.0 / (9 * cin) ** 0.5)
    bias = torch.randn(cout, device=dev) if bias_on else None
    xa = Act(x, 0, cin, sc, sh, 0.2)
    y = torch.empty(n, cout, h, w, device=dev)
    want = F.conv2d(act64(x, sc, sh, 0.2), wt.double(), None if bias is None else bias.double(), padding=1)
    # data gradient of the same layer: dy [n, cout, h, w] with an amax record -> dx [n, cin, h, w]
    dy = torch.randn(n, cout, h, w, device=dev) * 3e-5
    rec = ops.AMAX.next(dev)
    rec.zero_()
    rec.view(torch.float32)[0] = dy.abs().max()
    da = Act(dy, 0, cout)
    da.amax = rec
    dx = torch.empty(n, cin, h, w, device=dev)
    wantd = F.conv2d(dy.double(), wt.double().flip(2, 3).transpose(0, 1), padding=1)
    res = {}
    for on in (True, False):
        ops.conv3x3_wholek(on)
        for _ in range(REPS):
            part = ops.conv2d(xa, wt, bias, ops.full(y), stats=True, tag="w")
        for _ in range(REPS):
            ops.conv2d_dgrad(da, wt, ops.full(dx))
        torch.cuda.synchronize()
        e = ((y.double() - want).abs().max() / want.abs().max()).item()
        es = 0.0
        if part is not None:
            mu, var, tot = stats_from(part)
            assert float((tot - h * w).abs().max()) == 0.0, tot
            es = max(((mu - want.mean((2, 3))).abs().max() / want.abs().max()).item(),
                     ((var - want.var((2, 3), unbiased=False)).abs().max() / want.var((2, 3), unbiased=False).max()).item())
        ed = ((dx.double() - wantd).abs().max() / wantd.abs().max()).item()
        res[on] = (e, es, ed)
    worst = max(worst, *res[True])
    ops.conv3x3_wholek(True)
    print(f"conv {cin:3d}->{cout:3d} N {n} {h}x{w}: whole-K err {res[True][0]:.1e} stats {res[True][1]:.1e} dgrad {res[True][2]:.1e} | tiled {res[False][0]:.1e} {res[False][1]:.1e} {res[False][2]:.1e}", flush=True)
print("worst", worst)
assert worst < 3e-6
