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
for cin, cout, s in [(36,72,80),(72,72,80),(144,72,80),(72,144,40),(144,144,40),(288,144,40),(144,288,20),(288,288,20),(96,32,320),(64,64,160),(128,64,160),(72,36,160)]:
    x = torch.randn(8, cin, s, s, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    sc = torch.rand(8, cin, device=dev) + 0.5; sh = torch.randn(8, cin, device=dev)
    y = torch.empty(8, cout, s, s, device=dev); xa = ops.Act(x, 0, cin, sc, sh, 0.2); ya = ops.full(y)
    t = bench(lambda: ops.conv2d(xa, wt, None, ya, stats=True))
    fl = 2.0 * 8 * s * s * cin * cout * 9
    print(f"conv3 {cin:3d}->{cout:3d} @{s:3d}: {t:8.1f} us  {fl/t/1e6:6.1f} TF", flush=True)
