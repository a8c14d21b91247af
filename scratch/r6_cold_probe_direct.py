"""Cold-operand probe of the direct fp32 convolution (the cascade's 3->18 input conv, its 18->3 data gradient, the 18->2 output conv):
COLD=none|x (a different input per launch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = "cuda:0"
kind = os.environ.get("KIND", "dgrad")
cold = os.environ.get("COLD", "none")
N, s = 8, 320
if kind == "dgrad":       # dL/dx of Conv2d(3, 18, 3): dy has 18 channels
    cin, cout, ks = 3, 18, 3
elif kind == "fwd":
    cin, cout, ks = 3, 18, 3
else:                     # the 1x1 output conv 18 -> 2
    cin, cout, ks = 18, 2, 1
src_c = cout if kind == "dgrad" else cin
nx = max(2, int(400e6 / (N * src_c * s * s * 4))) if cold != "none" else 1
xs = [torch.randn(N, src_c, s, s, device=dev) for _ in range(nx)]
wt = torch.randn(cout, cin, ks, ks, device=dev) * 0.1
sc = torch.rand(N, src_c, device=dev) + 0.5
sh = torch.randn(N, src_c, device=dev) * 0.1
out = torch.empty(N, cin if kind == "dgrad" else cout, s, s, device=dev)
for i in range(200 + 2 * nx):
    x = xs[i % nx]
    if kind == "dgrad":
        ops.conv2d_dgrad(ops.full(x), wt, ops.full(out))
    else:
        ops.conv2d(ops.Act(x, 0, src_c, sc, sh, 0.2), wt, None, ops.full(out), stats=(kind == "fwd"))
torch.cuda.synchronize()
