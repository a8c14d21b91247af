import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = 'cuda:0'; N = 8
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cin, cout, s in [(18,18,320),(36,18,320),(18,36,160),(36,36,160),(72,36,160),(36,72,80),(72,72,80),(72,144,40)]:
    x = torch.randn(N, cin, s, s, device=dev); dy = torch.randn(N, cout, s, s, device=dev)
    dw = torch.empty(cout, cin, 3, 3, device=dev)
    xa = ops.full(x); da = ops.full(dy)
    os.environ['SAN_WGRAD_PRINT'] = '1'; ops.conv2d_wgrad(xa, da, dw); os.environ.pop('SAN_WGRAD_PRINT')
    res = []
    for rep in range(2):
        for mode in ('split', 'uniform'):
            if mode == 'uniform': os.environ['SAN_WGRAD_UNIFORM'] = '1'
            else: os.environ.pop('SAN_WGRAD_UNIFORM', None)
            res.append((mode, bench(lambda: ops.conv2d_wgrad(xa, da, dw))))
    os.environ.pop('SAN_WGRAD_UNIFORM', None)
    print((cin, cout, s), ' '.join(f"{m}={t:.1f}" for m, t in res), flush=True)
