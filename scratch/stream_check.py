"""Round 4: the stream (persistent) 3x3 convolution against float64 and against the one-tile-per-workgroup kernel, + timings.
    python scratch/stream_check.py            (SAN_CONV_STREAM=0 in the environment: the old kernel only)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops  # noqa: E402
from spatialalignmentnetwork_amd._lib import lib  # noqa: E402

dev = "cuda:0"
N = int(os.environ.get("BL_N", "8"))
LAYERS = [(18, 18, 320), (36, 18, 320), (18, 36, 320), (18, 36, 160), (36, 36, 160), (72, 36, 160), (36, 18, 160), (64, 32, 160), (48, 48, 160)]
if os.environ.get("BL_ONLY"):
    LAYERS = [tuple(int(v) for v in t.split("-")) for t in os.environ["BL_ONLY"].split(",")]


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


def ref64(x, sc, sh, slope, wt, bias):
    a = x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
    a = torch.where(a >= 0, a, a * slope)
    return torch.nn.functional.conv2d(a, wt.double(), None if bias is None else bias.double(), padding=1)


torch.manual_seed(0)
check = os.environ.get("SC_CHECK", "1") == "1"
for cin, cout, s in LAYERS:
    h = s
    w = s
    x = torch.randn(N, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    sc = torch.rand(N, cin, device=dev) + 0.5
    sh = torch.randn(N, cin, device=dev)
    y = torch.full((N, cout, h, w), float("nan"), device=dev)
    xa = ops.Act(x, 0, cin, sc, sh, 0.2)
    part = ops.conv2d(xa, wt, None, ops.full(y), stats=True)
    torch.cuda.synchronize()
    msg = ""
    if check:
        n_chk = min(N, 2)
        want = ref64(x[:n_chk], sc[:n_chk], sh[:n_chk], 0.2, wt, None)
        err = ((y[:n_chk].double() - want).norm() / want.norm()).item()
        # statistics: merge the per-wave records (count, mean, M2) and compare with the plane's mean / variance
        cnt, mean, m2 = part[..., 0].double(), part[..., 1].double(), part[..., 2].double()
        tot = cnt.sum(-1)
        mu = (cnt * mean).sum(-1) / tot
        var = (m2 + cnt * (mean - mu[..., None]) ** 2).sum(-1) / tot
        yd = y.double()
        e_mu = (mu - yd.mean((2, 3))).abs().max().item()
        e_var = ((var - yd.var((2, 3), unbiased=False)).abs() / yd.var((2, 3), unbiased=False)).max().item()
        full = (y.double()[n_chk:] - ref64(x[n_chk:], sc[n_chk:], sh[n_chk:], 0.2, wt, None)).abs().max().item() if N > n_chk else 0.0
        msg = f" rel-L2 {err:.2e} max-abs(rest) {full:.2e} stats mean {e_mu:.1e} var {e_var:.1e} nan {int(torch.isnan(y).sum())}"
    t = bench(lambda: ops.conv2d(xa, wt, None, ops.full(y), stats=True))
    fl = 2.0 * N * h * w * cin * cout * 9
    print(f"conv {cin:3d}->{cout:3d} @{s:3d}: {t:8.1f} us {fl / t / 1e6:6.1f} TF{msg}", flush=True)
    # data-gradient form: no affine, amax-scaled fp16 parts
    g = torch.randn(N, cout, h, w, device=dev) * 1e-6
    ga = ops.full(g)
    if ops.F16_BWD[0]:
        ga.amax = ops.amax_record(g.abs().max())
    dx = torch.full((N, cin, h, w), float("nan"), device=dev)
    ops.conv2d_dgrad(ga, wt, ops.full(dx))
    torch.cuda.synchronize()
    if check:
        want = torch.nn.functional.conv_transpose2d(g[:2].double(), wt.double(), padding=1)
        err = ((dx[:2].double() - want).norm() / want.norm()).item()
        msg = f" rel-L2 {err:.2e} nan {int(torch.isnan(dx).sum())}"
    t = bench(lambda: ops.conv2d_dgrad(ga, wt, ops.full(dx)))
    print(f"dgrad {cout:3d}->{cin:3d} @{s:3d}: {t:8.1f} us {fl / t / 1e6:6.1f} TF{msg}", flush=True)
