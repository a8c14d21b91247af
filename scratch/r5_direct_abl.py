import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatialalignmentnetwork_amd import ops
from spatialalignmentnetwork_amd.ops import Act
dev = "cuda:0"
n, cin, cout, h, w = 8, 4, 18, 320, 320
x = torch.randn(n, cin, h, w, device=dev); sc = torch.rand(n, cin, device=dev) + 0.5; sh = torch.randn(n, cin, device=dev) * .3
wt = torch.randn(cout, cin, 3, 3, device=dev) * .1
y = torch.empty(n, cout, h, w, device=dev)
xa = Act(x, 0, cin, sc, sh, 0.2)
for mode in (1, 2, 3):
    ops.lib().query("san_conv_direct_enable", mode); ops.lib()._memo.clear()
    for st in (True, False):
        for _ in range(20):
            ops.conv2d(xa, wt, None, ops.full(y), stats=st, tag="t")
        torch.cuda.synchronize()
