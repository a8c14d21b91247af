"""1x1 weight gradient: bf16x3 kernel vs the fp32 kernel at the bench workload's shapes (N=8)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = 'cuda:0'
def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cin, cout, s in [(288, 576, 20), (144, 288, 40), (72, 144, 80), (36, 72, 160), (32, 64, 160), (64, 64, 160), (64, 64, 80), (64, 64, 40), (64, 64, 20), (18, 2, 320), (8, 2, 320), (32, 2, 320)]:
    N = 8
    x = torch.randn(N, cin, s, s, device=dev); dy = torch.randn(N, cout, s, s, device=dev)
    sc = torch.rand(N, cin, device=dev) + 0.5; sh = torch.randn(N, cin, device=dev)
    xa = ops.Act(x, 0, cin, sc, sh, 0.2); da = ops.full(dy)
    dw = torch.empty(cout, cin, 1, 1, device=dev)
    msg = f"1x1 {cin:3d}->{cout:3d} @{s:3d}:"
    act = torch.nn.functional.leaky_relu(x * sc[:, :, None, None] + sh[:, :, None, None], 0.2).double()
    ref = torch.einsum("nohw,nihw->oi", dy.double(), act)[:, :, None, None]
    for name in ("bf16x3", "fp32"):
        ops.USE_BF16X3[0] = name == "bf16x3"
        fn = ops.conv2d_wgrad1x1_bf16x3 if name == "bf16x3" else ops.conv2d_wgrad
        fn(xa, da, dw)
        err = ((dw.double() - ref).norm() / ref.norm()).item()
        t = bench(lambda: fn(xa, da, dw))
        msg += f"  {name} {t:7.1f} us rel {err:.1e}"
    ops.USE_BF16X3[0] = True
    print(msg, flush=True)
