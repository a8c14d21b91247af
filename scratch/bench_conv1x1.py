"""1x1 convolution: bf16x3 kernel vs the fp32 kernel (forward with stats, and data gradient), N=8."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = 'cuda:0'
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm()).item()
for cin, cout, s in [(576, 288, 20), (288, 144, 40), (144, 72, 80), (72, 36, 160), (32, 64, 160), (64, 64, 160), (64, 64, 80), (64, 64, 40), (128, 64, 160), (18, 2, 320), (36, 72, 160)]:
    N = 8
    x = torch.randn(N, cin + 1, s, s, device=dev); wt = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
    sc = torch.rand(N, cin + 1, device=dev) + 0.5; sh = torch.randn(N, cin + 1, device=dev); b = torch.randn(cout, device=dev)
    y = torch.empty(N, cout, s, s, device=dev)
    xa = ops.Act(x, 1, cin, sc, sh, 0.2); ya = ops.full(y)
    act = torch.nn.functional.leaky_relu(x[:, 1:] * sc[:, 1:, None, None] + sh[:, 1:, None, None], 0.2)
    ref = torch.nn.functional.conv2d(act.double(), wt.double(), b.double())
    msg = f"1x1 {cin:3d}->{cout:3d} @{s:3d}:"
    for name in ("bf16x3", "fp32"):
        ops.USE_BF16X3[0] = name == "bf16x3"
        if name == "bf16x3" and not ops.bf16x3_eligible(cin, cout, s, s, 1):
            msg += "  bf16x3 (not eligible)"; continue
        part = ops.conv2d(xa, wt, b, ya, stats=True)
        e = rel(y, ref)
        p = part.double(); cnt, mean_t = p[..., 0], p[..., 1]
        mean = (cnt * mean_t).sum(-1) / cnt.sum(-1)
        em = (mean - ref.mean(dim=(2, 3))).abs().max().item()
        t = bench(lambda: ops.conv2d(xa, wt, b, ya, stats=True))
        msg += f"  {name} {t:7.1f} us rel {e:.1e} mean {em:.1e}"
    ops.USE_BF16X3[0] = True
    print(msg, flush=True)
