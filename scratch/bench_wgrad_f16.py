"""Per-layer timing of the weight-gradient kernel at the bench workload's layer shapes (N=8)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = 'cuda:0'; N = 8
def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
shapes = [(18,18,320,3),(36,18,320,3),(18,36,160,3),(36,36,160,3),(72,36,160,3),(36,72,80,3),(72,72,80,3),
          (144,72,80,3),(72,144,40,3),(144,144,40,3),(288,144,40,3),(144,288,20,3),(288,288,20,3),(18,2,320,1),(288,576,20,1),(36,72,160,1)]
check = '--check' in sys.argv
tot = 0.0
for cin, cout, s, ks in shapes:
    x = torch.randn(N, cin, s, s, device=dev); dy = torch.randn(N, cout, s, s, device=dev) * 1e-5
    sc = torch.rand(N, cin, device=dev) + 0.5; sh = torch.randn(N, cin, device=dev)
    dw = torch.empty(cout, cin, ks, ks, device=dev)
    xa = ops.Act(x, 0, cin, sc, sh, 0.2); da = ops.full(dy); da.amax = ops.amax_record(dy.abs().max()) if ops.F16_BWD[0] else None
    t = bench(lambda: ops.conv2d_wgrad(xa, da, dw))
    fl = 2.0 * N * s * s * cin * cout * ks * ks
    msg = f"cin={cin:3d} cout={cout:3d} {s:3d}^2 k{ks}: {t:8.1f} us  {fl / t / 1e6:6.1f} TF"
    if check:
        act = torch.nn.functional.leaky_relu(x * sc[:, :, None, None] + sh[:, :, None, None], 0.2).double()
        w = torch.zeros(cout, cin, ks, ks, device=dev, dtype=torch.double, requires_grad=True)
        torch.nn.functional.conv2d(act, w, padding=ks // 2).backward(dy.double())
        err = ((dw.double() - w.grad).norm() / w.grad.norm()).item()
        msg += f"  rel_err={err:.2e}"
    print(msg, flush=True)
    if ks == 3: tot += t
print(f"sum 3x3: {tot:.0f} us")
