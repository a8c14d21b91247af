"""Cold-operand probe of the weight-gradient kernels (see r6_cold_probe.py): COLD=none|xdy (a different x / dy pair per launch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = "cuda:0"
cin, cout, s, N = [int(v) for v in os.environ.get("LAYER", "18-18-320-8").split("-")]
cold = os.environ.get("COLD", "none")
nx = max(2, int(400e6 / (N * (cin + cout) * s * s * 4))) if cold != "none" else 1
xs = [torch.randn(N, cin, s, s, device=dev) for _ in range(nx)]
dys = [torch.randn(N, cout, s, s, device=dev) * 1e-3 for _ in range(nx)]
sc = torch.rand(N, cin, device=dev) + 0.5
sh = torch.randn(N, cin, device=dev)
dw = torch.zeros(cout, cin, 3, 3, device=dev)
am = ops.amax_record(dys[0].abs().max() * 4)
print(cold, "pairs", nx, flush=True)
for i in range(200 + 2 * nx):
    dya = ops.full(dys[i % nx]); dya.amax = am
    ops.conv2d_wgrad(ops.Act(xs[i % nx], 0, cin, sc, sh, 0.2), dya, dw, accumulate=False)
torch.cuda.synchronize()
