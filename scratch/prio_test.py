import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from spatialalignmentnetwork_amd import synth
dev = torch.device('cuda', 0)
n, h, w = 8, 320, 320
net = bench.build_model(n, h, w, 12, dev)
a, b = synth.phantom_pair(n, 1, h, w, seed=1234)
a, b = a.to(dev), b.to(dev)
net.train()
def run(stream):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(3): bench.train_step(net, a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): bench.train_step(net, a, b)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10 * 1e3
hi = torch.cuda.Stream(priority=-1)
for i in range(2):
    print("main default:", round(run(None), 2), " main high-priority:", round(run(hi), 2), flush=True)
