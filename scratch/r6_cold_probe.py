"""Why is a 40^2-level convolution 30-40 % slower inside the step than alone?  One layer, launched 200 times, with
  COLD=none     the same input and weights every time (everything L2-hot: what scratch/bench_layers.py measures)
  COLD=w        a different weight image every launch (cycling through > 256 MB of packed weights: HBM-cold weights)
  COLD=x        a different input tensor every launch (cycling through > 256 MB of activations)
  COLD=wx       both
  COLD=prod     input written by another kernel right before (as in the step: produced on other XCDs, L2-cold but MALL-warm), weights cold
Timed by rocprofv3 --kernel-trace --stats around this process (scratch/r6_cold_probe.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = "cuda:0"
cin, cout, s, N = [int(v) for v in os.environ.get("LAYER", "144-144-40-8").split("-")]
cold = os.environ.get("COLD", "none")
nw = max(1, int(320e6 / (cin * cout * 9 * 12))) if "w" in cold or cold == "prod" else 1
nx = max(1, int(320e6 / (N * cin * s * s * 4))) if "x" in cold else 1
ws = [torch.randn(cout, cin, 3, 3, device=dev) * 0.05 for _ in range(nw)]
xs = [torch.randn(N, cin, s, s, device=dev) for _ in range(nx)]
sc = torch.rand(N, cin, device=dev) + 0.5
sh = torch.randn(N, cin, device=dev)
y = torch.empty(N, cout, s, s, device=dev)
print(cold, "weights", nw, "inputs", nx, flush=True)
for i in range(200 + 2 * max(nw, nx)):
    x = xs[i % nx]
    if cold == "prod":
        x.mul_(1.0)                      # (re-written by another kernel: lines leave this XCD's L2 through the write-back)
    ops.conv2d(ops.Act(x, 0, cin, sc, sh, 0.2), ws[i % nw], None, ops.full(y), stats=True)
torch.cuda.synchronize()
