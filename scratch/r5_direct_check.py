"""Direct small-channel convolution (round 5) vs float64 and vs the outer-product kernel; timings at the cascade's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from spatialalignmentnetwork_amd import ops
from spatialalignmentnetwork_amd.ops import Act
dev = "cuda:0"
torch.manual_seed(0)


def ref64(x, sc, sh, slope, wt, bias):
    xd = x.double()
    if sc is not None:
        xd = xd * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
        xd = torch.where(xd >= 0, xd, xd * slope)
    return F.conv2d(xd, wt.double(), None if bias is None else bias.double(), padding=wt.shape[-1] // 2)


def bench(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def stats_from(part):
    cnt, mean, m2 = part[..., 0].double(), part[..., 1].double(), part[..., 2].double()
    tot = cnt.sum(-1)
    mu = (cnt * mean).sum(-1) / tot
    var = (m2 + cnt * (mean - mu[..., None]) ** 2).sum(-1) / tot
    return mu, var


worst = 0.0
for (n, cin, cout, h, w, ks, dgrad, aff, bias_on) in [
        (8, 4, 18, 320, 320, 3, False, True, False), (8, 18, 2, 320, 320, 1, False, True, True),
        (8, 2, 18, 320, 320, 1, True, False, False), (8, 18, 4, 320, 320, 3, True, False, False),
        (2, 3, 7, 50, 37, 3, False, True, True), (1, 2, 64, 96, 132, 3, False, False, True), (3, 20, 3, 61, 70, 3, True, False, False),
        (2, 1, 8, 40, 23, 1, False, True, False), (1, 36, 2, 160, 92, 1, False, True, True), (15, 2, 8, 640, 368, 3, False, False, False)]:
    x = torch.randn(n, cin, h, w, device=dev)
    sc = (torch.rand(n, cin, device=dev) + 0.5) if aff else None
    sh = (torch.randn(n, cin, device=dev) * 0.3) if aff else None
    if dgrad:
        wt = torch.randn(cin, cout, ks, ks, device=dev) * 0.1        # forward weight [cout_f = cin here][cin_f = cout here]
    else:
        wt = torch.randn(cout, cin, ks, ks, device=dev) * 0.1
    bias = torch.randn(cout, device=dev) if bias_on else None
    y = torch.empty(n, cout, h, w, device=dev)
    xa = Act(x, 0, cin, sc, sh, 0.2 if aff else 1.0)
    res = {}
    for on in (True, False):
        ops.conv_direct(on)
        if dgrad:
            fn = lambda: ops.conv2d_dgrad(xa, wt, ops.full(y))
            fn()
            part = None
        else:
            fn = lambda: ops.conv2d(xa, wt, bias, ops.full(y), stats=True, tag="t")
            part = fn().clone()
        torch.cuda.synchronize()
        res[on] = (y.clone(), part, bench(fn))
    ops.conv_direct(True)
    if dgrad:
        want = ref64(x, sc, sh, 0.2, wt.flip(2, 3).transpose(0, 1), None)
    else:
        want = ref64(x, sc, sh, 0.2, wt, bias)
    e_d = ((res[True][0].double() - want).abs().max() / want.abs().max()).item()
    e_m = ((res[False][0].double() - want).abs().max() / want.abs().max()).item()
    es = 0.0
    if not dgrad:
        mu, var = stats_from(res[True][1])
        es = max(((mu - want.mean((2, 3))).abs().max() / want.abs().max()).item(), ((var - want.var((2, 3), unbiased=False)).abs().max() / want.var((2, 3), unbiased=False).max()).item())
    worst = max(worst, e_d, es)
    print(f"{'dgrad' if dgrad else 'conv '} {cin:3d}->{cout:3d} ks {ks} N {n:2d} {h}x{w}: direct {res[True][2]:6.1f} us (err {e_d:.1e}, stats {es:.1e})   outer-product {res[False][2]:6.1f} us (err {e_m:.1e})", flush=True)
print("worst", worst)
assert worst < 3e-6
