"""Workload for the HBM-traffic PMC passes: a calibration copy of known size, then train steps."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import synth
dev = torch.device('cuda', 0)
# calibration kernels of known traffic, each far beyond the 256 MiB Infinity Cache:
#   apply_kernel  16 B/lane: [8,64,512*512] fp32  -> 512 MiB read, 512 MiB written
#   apply_kernel   4 B/lane: [8,64,511*513] fp32  -> 511.998 MiB read / written
#   rss_kernel     8 B/lane: [32,8,512,512] c64   -> 512 MiB read, 32 MiB written
from spatialalignmentnetwork_amd import ops
for hw_shape in ((512, 512), (511, 513)):
    a = torch.randn(8, 64, *hw_shape, device=dev); b = torch.empty_like(a)
    for _ in range(2): ops.apply(ops.full(a), ops.full(b))
    del a, b
c = torch.randn(32, 8, 512, 512, dtype=torch.complex64, device=dev)
for _ in range(2): ops.rss(c)
del c
torch.cuda.synchronize()
n, h, w = 8, 320, 320
net = bench.build_model(n, h, w, 12, dev)
x, y = synth.phantom_pair(n, 1, h, w, seed=1234)
x, y = x.to(dev), y.to(dev)
net.train()
for _ in range(3): bench.train_step(net, x, y)
torch.cuda.synchronize()
