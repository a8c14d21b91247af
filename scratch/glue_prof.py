"""Which Python lines launch the small ATen kernels of a train step (torch.profiler with stacks)."""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import synth
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
n, h, w = 8, 320, 320
net = bench.build_model(n, h, w, 12, dev)
a, b = synth.phantom_pair(n, 1, h, w, seed=1234)
a, b = a.to(dev), b.to(dev)
net.train()
for _ in range(3): bench.train_step(net, a, b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.train_step(net, a, b)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::"): continue
    dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"): continue     # count top-level ops only
    frame = next((s for s in ev.stack if "spatialalignmentnetwork_amd" in s or "bench.py" in s), "?")
    key = (frame.split("spatialalignmentnetwork_amd/")[-1][:70], ev.name)
    agg[key][0] += 1; agg[key][1] += dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
print("top-level aten ops per step:", tot)
for (frame, name), (c, dt) in rows[:45]:
    print(f"{c:5d}  {dt / 1e3:7.3f} ms  {name:28s} {frame}")
