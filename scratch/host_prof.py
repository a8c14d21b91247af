"""cProfile of the host side of one training step (where does the enqueue time go?): python scratch/host_prof.py"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import synth
dev = torch.device('cuda', 0)
n, h, w = 8, 320, 320
net = bench.build_model(n, h, w, 12, dev)
a, b = synth.phantom_pair(n, 1, h, w, seed=1234)
a, b = a.to(dev), b.to(dev)
net.train()
for _ in range(4): bench.train_step(net, a, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): bench.train_step(net, a, b)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/5:.1f} ms/step, total {1e3*(t2-t0)/5:.1f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3): bench.train_step(net, a, b)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45); st.sort_stats("cumulative").print_stats(60)
