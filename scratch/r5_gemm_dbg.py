import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from spatialalignmentnetwork_amd import ops
from spatialalignmentnetwork_amd.ops import Act
dev = "cuda:0"
torch.manual_seed(0)
n, cin, cout, h, w = 2, 16, 8, 16, 32
x = torch.randn(n, cin, h, w, device=dev)
sc, sh = torch.ones(n, cin, device=dev), torch.zeros(n, cin, device=dev)
wt = torch.randn(cin, cout, 2, 2, device=dev) * 0.25
xa = Act(x, 0, cin, sc, sh, 1.0)
y = torch.empty(n, cout, 2 * h, 2 * w, device=dev)
for on in (True, False):
    ops.conv1x1_gemm(on)
    part = ops.tconv2x2(xa, wt, ops.full(y), stats=True, tag="t").clone()
    torch.cuda.synchronize()
    print("gemm", on, "part shape", tuple(part.shape))
    print(part[0, 0, :, :].cpu())
    print(part[1, 5, :12, :].cpu())
