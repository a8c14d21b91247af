import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import synth
dev = torch.device('cuda', 0)
n, h, w = 8, 320, 320
net = bench.build_model(n, h, w, 12, dev)
a, b = synth.phantom_pair(n, 1, h, w, seed=1234)
a, b = a.to(dev), b.to(dev)
net.eval()
for _ in range(3): bench.one_step(net, a, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): bench.one_step(net, a, b)
torch.cuda.synchronize()
print("wall ms/step", 1e3 * (time.perf_counter() - t0) / 5)
