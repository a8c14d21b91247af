"""Which C-ABI entry points does one training step call, how often (host side)?  python scratch/count_calls.py
Also lists the individually packed weights (PACKS16._pack_one) and ATen kernels are visible in rocprof only."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import ops, synth, _lib
dev = torch.device('cuda', 0)
n, h, w = 8, 320, 320
net = bench.build_model(n, h, w, 12, dev)
a, b = synth.phantom_pair(n, 1, h, w, seed=1234)
a, b = a.to(dev), b.to(dev)
net.train()
for _ in range(3): bench.train_step(net, a, b)
torch.cuda.synchronize()
cnt = collections.Counter()
L = _lib.lib()
orig = L.call
def call(name, *args):
    cnt[name] += 1
    return orig(name, *args)
L.call = call
packs = collections.Counter()
o1 = ops.PACKS16._pack_one
def p1(job, w):
    packs[(tuple(w.shape), job["mode"])] += 1
    return o1(job, w)
ops.PACKS16._pack_one = p1
regs = collections.Counter()
o2 = ops.PACKS16._register
def r2(w, mode):
    regs[(tuple(w.shape), mode)] += 1
    return o2(w, mode)
ops.PACKS16._register = r2
bench.train_step(net, a, b)
torch.cuda.synchronize()
print("C-ABI calls in one step:", sum(cnt.values()))
for k, v in cnt.most_common(): print(f"  {v:5d}  {k}")
print("individually packed weights:", sum(packs.values()), dict(packs))
print("re-registered weights:", sum(regs.values()), dict(regs))
