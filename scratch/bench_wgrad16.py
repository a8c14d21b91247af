"""bf16x3 weight gradient vs the fp32 kernel: accuracy against float64 (--check) and per-layer time (N=8)."""
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
def run(N, cin, cout, h, w, check, time=True):
    x = torch.randn(N, cin, h, w, device=dev); dy = torch.randn(N, cout, h, w, device=dev)
    sc = torch.rand(N, cin, device=dev) + 0.5; sh = torch.randn(N, cin, device=dev)
    xa = ops.Act(x, 0, cin, sc, sh, 0.2); da = ops.full(dy)
    out = {}
    msg = f"n={N} {cin:3d}->{cout:3d} {h:3d}x{w:3d}:"
    for name, flag in (("split", True), ("direct", True), ("fp32", False)):
        ops.USE_BF16X3[0] = flag
        ops.lib().call("san_conv_wgrad_bf16x3_set_mode", {"split": 0, "direct": 1, "fp32": -1}[name])
        dw = torch.full((cout, cin, 3, 3), float('nan'), device=dev)
        fn = ops.conv2d_wgrad_bf16x3 if flag else ops.conv2d_wgrad
        fn(xa, da, dw)
        out[name] = dw.clone()
        if time:
            t = bench(lambda: fn(xa, da, dw))
            msg += f"  {name} {t:7.1f} us ({2.0 * N * h * w * cin * cout * 9 / t / 1e6:5.1f} TF)"
    ops.USE_BF16X3[0] = True
    if check:
        act = torch.nn.functional.leaky_relu(x * sc[:, :, None, None] + sh[:, :, None, None], 0.2).double()
        wt = torch.zeros(cout, cin, 3, 3, device=dev, dtype=torch.double, requires_grad=True)
        torch.nn.functional.conv2d(act, wt, padding=1).backward(dy.double())
        for name in out:
            err = ((out[name].double() - wt.grad).norm() / wt.grad.norm()).item()
            mx = ((out[name].double() - wt.grad).abs().max() / wt.grad.abs().max()).item()
            msg += f"  {name} rel {err:.2e} max {mx:.2e}"
        # accumulate path
        ops.lib().call("san_conv_wgrad_bf16x3_set_mode", 1)
        dw = out["direct"].clone()
        ops.conv2d_wgrad_bf16x3(xa, da, dw, accumulate=True)
        msg += f"  acc {((dw.double() - 2 * wt.grad).norm() / wt.grad.norm()).item():.1e}"
    print(msg, flush=True)
if '--check' in sys.argv:
    for N, cin, cout, h, w in [(2, 32, 32, 16, 32), (1, 72, 72, 20, 44), (2, 36, 72, 40, 40), (1, 96, 32, 33, 52), (2, 144, 144, 24, 24),
                               (1, 288, 144, 16, 16), (3, 40, 50, 7, 12), (1, 32, 48, 23, 23), (2, 64, 64, 30, 46)]:
        run(N, cin, cout, h, w, True, time=False)
for cin, cout, s in [(3, 18, 320), (4, 18, 320), (8, 8, 320), (8, 16, 160), (16, 16, 160), (16, 32, 80), (16, 8, 320), (18, 18, 320), (36, 18, 320), (18, 36, 160), (32, 16, 320), (24, 24, 160), (36, 36, 160), (72, 36, 160), (36, 72, 80), (72, 72, 80), (144, 72, 80), (72, 144, 40), (144, 144, 40), (288, 144, 40),
                     (144, 288, 20), (288, 288, 20), (32, 32, 320), (96, 32, 320), (64, 64, 160), (128, 64, 160), (64, 64, 80), (64, 64, 40)]:
    run(8, cin, cout, s, s, False)
