"""18 -> 3 @320^2 data gradient on the persistent matrix-core kernel (partial block alone) against the direct fp32 kernel: error vs
float64 and time (SAN_STREAM_SMALL_COUT=1 must be set for the new path)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
import torch.nn.functional as F
dev = "cuda:0"
torch.manual_seed(0)
n, cin, cout, h, w = 8, 3, 18, 320, 320
wt = (torch.randn(cout, cin, 3, 3, device=dev) * 0.1)
g = torch.randn(n, cout, h, w, device=dev) * 1e-3
ops.AMAX.reset(dev)
dy = ops.Act(torch.empty_like(g), 0, cout)
ops.act_bwd(ops.full(g), ops.full(torch.ones_like(g)), dy, instance_norm=False)
dx = torch.empty(n, cin, h, w, device=dev)
ops.conv2d_dgrad(dy, wt, ops.full(dx))
torch.cuda.synchronize()
ref = F.conv_transpose2d(g.double(), wt.double(), padding=1)
print("small path on:", ops.STREAM_SMALL_COUT[0], " rel err vs float64: %.2e" % ((dx.double() - ref).norm() / ref.norm()).item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): ops.conv2d_dgrad(dy, wt, ops.full(dx))
e0.record()
for _ in range(50): ops.conv2d_dgrad(dy, wt, ops.full(dx))
e1.record(); torch.cuda.synchronize()
print("us per launch (hot): %.1f" % (e0.elapsed_time(e1) * 1e3 / 50))
