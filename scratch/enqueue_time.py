"""Host time to ENQUEUE one train step vs its GPU time (is the step launch-bound?)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import synth
dev = torch.device('cuda', 0)
n, h, w = 8, 320, 320
net = bench.build_model(n, h, w, 12, dev)
a, b = synth.phantom_pair(n, 1, h, w, seed=1234)
a, b = a.to(dev), b.to(dev)
net.train()
for _ in range(3): bench.train_step(net, a, b)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(5):
    t0 = time.perf_counter()
    bench.train_step(net, a, b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(1e3 * (t1 - t0)); tot.append(1e3 * (t2 - t0))
print("enqueue ms", [f"{x:.1f}" for x in enq], "total ms", [f"{x:.1f}" for x in tot])
