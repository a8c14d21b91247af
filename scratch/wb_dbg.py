import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev='cuda:0'
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cin, cout, s in [(72,72,80),(144,72,80),(144,144,40),(288,288,20),(64,64,160),(128,64,160),(96,32,320)]:
    x = torch.randn(8, cin, s, s, device=dev); dy = torch.randn(8, cout, s, s, device=dev)
    dw = torch.empty(cout, cin, 3, 3, device=dev)
    xa = ops.full(x); da = ops.full(dy)
    print(cin, cout, s, os.environ.get("SAN_WB_DBG"), f"{bench(lambda: ops.conv2d_wgrad_bf16x3(xa, da, dw)):.1f} us")
