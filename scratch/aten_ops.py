"""Which ATen ops does one training step (set_input + update) still dispatch?  python scratch/aten_ops.py"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import synth
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda', 0)
n, h, w = 8, 320, 320
net = bench.build_model(n, h, w, 12, dev)
a, b = synth.phantom_pair(n, 1, h, w, seed=1234)
a, b = a.to(dev), b.to(dev)
net.train()
for _ in range(3): bench.train_step(net, a, b)
torch.cuda.synchronize()
cnt = collections.Counter()
import traceback
where = collections.defaultdict(collections.Counter)
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        cnt[name] += 1
        for fr in reversed(traceback.extract_stack(limit=12)):
            if 'spatialalignmentnetwork_amd' in fr.filename and 'aten_ops' not in fr.filename:
                where[name][f"{os.path.basename(fr.filename)}:{fr.lineno}"] += 1
                break
        return func(*args, **(kwargs or {}))
with Log():
    bench.train_step(net, a, b)
torch.cuda.synchronize()
print("ATen dispatches in one step:", sum(cnt.values()))
for k, v in cnt.most_common():
    print(f"{v:5d} {k:40s} {dict(where[k].most_common(6))}")
