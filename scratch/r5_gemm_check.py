"""1x1 / transposed convolutions as one-stage GEMMs (round 5) vs float64 and vs the tiled kernel's KS = 1 form."""
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
    cnt, mean, m2 = part[..., 0].double(), part[..., 1].double(), part[..., 2].double()
    tot = cnt.sum(-1)
    mu = (cnt * mean).sum(-1) / tot
    var = (m2 + cnt * (mean - mu[..., None]) ** 2).sum(-1) / tot
    return mu, var, tot


worst = 0.0
for (n, cin, cout, h, w) in [(8, 288, 144, 20, 20), (8, 144, 72, 40, 40), (8, 72, 36, 80, 80), (8, 36, 18, 160, 160),
                             (2, 40, 20, 23, 46), (1, 64, 16, 92, 160), (15, 16, 8, 160, 92)]:
    x = torch.randn(n, cin, h, w, device=dev)
    sc, sh = torch.rand(n, cin, device=dev) + 0.5, torch.randn(n, cin, device=dev) * 0.3
    wt = torch.randn(cin, cout, 2, 2, device=dev) * (1.0 / cin ** 0.5)
    xa = Act(x, 0, cin, sc, sh, 0.2)
    y = torch.empty(n, cout, 2 * h, 2 * w, device=dev)
    want = F.conv_transpose2d(act64(x, sc, sh, 0.2), wt.double(), stride=2)
    res = {}
    for on in (True, False):
        ops.conv1x1_gemm(on)
        for _ in range(REPS):
            part = ops.tconv2x2(xa, wt, ops.full(y), stats=True, tag="t")
        torch.cuda.synchronize()
        mu, var, tot = stats_from(part)
        e = ((y.double() - want).abs().max() / want.abs().max()).item()
        es = max(((mu - want.mean((2, 3))).abs().max() / want.abs().max()).item(),
                 ((var - want.var((2, 3), unbiased=False)).abs().max() / want.var((2, 3), unbiased=False).max()).item())
        assert float((tot - 4 * h * w).abs().max()) == 0.0, tot
        res[on] = (e, es)
        worst = max(worst, e, es) if on else worst
    # data gradient: dy' [n, 4 cout, h, w] with an amax record -> dx [n, cin, h, w]
    dyp = torch.randn(n, 4 * cout, h, w, device=dev) * 3e-5
    rec = ops.AMAX.next(dev)
    rec.zero_()
    rec.view(torch.float32)[0] = dyp.abs().max()
    da = Act(dyp, 0, 4 * cout)
    da.amax = rec
    dx = torch.empty(n, cin, h, w, device=dev)
    wv = wt.reshape(cin, 4 * cout, 1, 1)
    wantd = F.conv2d(dyp.double(), wv.double())
    rd = {}
    for on in (True, False):
        ops.conv1x1_gemm(on)
        for _ in range(REPS):
            ops.conv2d(da, wv, None, ops.full(dx), grad_input=True)
        torch.cuda.synchronize()
        rd[on] = ((dx.double() - wantd).abs().max() / wantd.abs().max()).item()
    worst = max(worst, rd[True])
    ops.conv1x1_gemm(True)
    print(f"tconv {cin:3d}->{cout:3d} N {n:2d} {h}x{w}: gemm err {res[True][0]:.1e} stats {res[True][1]:.1e} | tiled {res[False][0]:.1e} {res[False][1]:.1e} || dgrad gemm {rd[True]:.1e} tiled {rd[False]:.1e}", flush=True)
# plain 1x1 with bias and statistics, channel views
for (n, cin, cout, h, w) in [(8, 64, 64, 160, 160), (2, 48, 40, 33, 50), (8, 128, 32, 80, 80)]:
    xb = torch.randn(n, cin + 5, h, w, device=dev)
    sc, sh = torch.rand(n, cin + 5, device=dev) + 0.5, torch.randn(n, cin + 5, device=dev) * 0.3
    wt = torch.randn(cout, cin, 1, 1, device=dev) * (1.0 / cin ** 0.5)
    bias = torch.randn(cout, device=dev)
    yb = torch.zeros(n, cout + 3, h, w, device=dev)
    xa = Act(xb, 2, cin, sc, sh, 0.2)
    want = F.conv2d(act64(xb[:, 2:2 + cin], sc[:, 2:2 + cin], sh[:, 2:2 + cin], 0.2), wt.double(), bias.double())
    for on in (True, False):
        ops.conv1x1_gemm(on)
        for _ in range(REPS):
            part = ops.conv2d(xa, wt, bias, Act(yb, 1, cout), stats=True, tag="u")
        torch.cuda.synchronize()
        mu, var, tot = stats_from(part)
        e = ((yb[:, 1:1 + cout].double() - want).abs().max() / want.abs().max()).item()
        es = max(((mu - want.mean((2, 3))).abs().max() / want.abs().max()).item(),
                 ((var - want.var((2, 3), unbiased=False)).abs().max() / want.var((2, 3), unbiased=False).max()).item())
        assert float(yb[:, 0].abs().max()) == 0.0 and float(yb[:, 1 + cout:].abs().max()) == 0.0
        print(f"1x1 {cin:3d}->{cout:3d} N {n} {h}x{w} gemm={on}: err {e:.1e} stats {es:.1e}", flush=True)
        worst = max(worst, e, es) if on else worst
ops.conv1x1_gemm(True)
print("worst", worst)
assert worst < 3e-6
