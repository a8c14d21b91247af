"""Stand-alone check of the bf16x3 conv kernel vs float64 conv (fwd with stats, bias, lazy affine; dgrad) + timing."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = 'cuda:0'
torch.manual_seed(0)
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm()).item()
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, cin, cout, h, w in [(2, 24, 32, 16, 32), (1, 72, 72, 20, 44), (2, 36, 72, 40, 40), (1, 96, 32, 33, 50), (2, 144, 144, 24, 24), (1, 288, 144, 16, 16)]:
    x = torch.randn(n, cin + 3, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    sc = torch.rand(n, cin + 3, device=dev) + 0.5; sh = torch.randn(n, cin + 3, device=dev); b = torch.randn(cout, device=dev)
    y = torch.empty(n, cout + 2, h, w, device=dev)
    xa = ops.Act(x, 3, cin, sc, sh, 0.2); ya = ops.Act(y, 2, cout, None, None, 1.0)
    part = ops.conv2d(xa, wt, b, ya, stats=True)
    act = torch.nn.functional.leaky_relu(x[:, 3:] * sc[:, 3:, None, None] + sh[:, 3:, None, None], 0.2).double()
    ref = torch.nn.functional.conv2d(act, wt.double(), b.double(), padding=1)
    e = rel(y[:, 2:], ref)
    # statistics: merge tiles (Chan) and compare with mean / biased var of ref
    cnt = part[..., 0].double(); mean_t = part[..., 1].double(); m2_t = part[..., 2].double()
    tot = cnt.sum(-1); mean = (cnt * mean_t).sum(-1) / tot
    m2 = (m2_t + cnt * (mean_t - mean[..., None]) ** 2).sum(-1)
    em = (mean - ref.mean(dim=(2, 3))).abs().max().item(); ev = rel(m2 / tot, ref.var(dim=(2, 3), unbiased=False))
    # dgrad
    dy = torch.randn(n, cout, h, w, device=dev); dx = torch.empty(n, cin, h, w, device=dev)
    ops.conv2d_dgrad(ops.full(dy), wt, ops.full(dx))
    a64 = act.clone().requires_grad_(True)
    torch.nn.functional.conv2d(a64, wt.double(), None, padding=1).backward(dy.double())
    ed = rel(dx, a64.grad)
    print(f"n={n} {cin}->{cout} {h}x{w}: fwd rel {e:.2e}  mean abs {em:.2e}  var rel {ev:.2e}  count {tot.min().item():.0f}/{h*w}  dgrad rel {ed:.2e}", flush=True)
print("timing (N=8):")
import ctypes
elig_real = ops.bf16x3_eligible
for cin, cout, s in [(3,18,320),(8,8,320),(8,16,160),(16,16,160),(16,32,80),(18,2,320),(24,24,160),(64,32,320),(32,64,160),(18,36,320),(18,36,160),(36,36,160),(72,36,160),(36,18,320),(18,18,320),(32,32,320),(36,72,80),(144,72,80),(288,288,20),(144,288,20),(32,2,320)]:
    x = torch.randn(8, cin, s, s, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    sc = torch.rand(8, cin, device=dev) + 0.5; sh = torch.randn(8, cin, device=dev)
    y = torch.empty(8, cout, s, s, device=dev); xa = ops.Act(x, 0, cin, sc, sh, 0.2); ya = ops.full(y)
    res = []
    for flag in (True, False):
        ops.bf16x3_eligible = (lambda *a, f=flag: f)
        res.append(bench(lambda: ops.conv2d(xa, wt, None, ya, stats=True)))
    ops.bf16x3_eligible = elig_real
    fl = 2.0 * 8 * s * s * cin * cout * 9
    print(f"conv3 {cin:3d}->{cout:3d} @{s:3d}: bf16x3 {res[0]:7.1f} us ({fl/res[0]/1e6:6.1f} TF)   fp32 {res[1]:7.1f} us ({fl/res[1]/1e6:6.1f} TF)", flush=True)
