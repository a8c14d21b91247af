import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import synth, ops
dev = torch.device('cuda', 0)
net = bench.build_model(8, 320, 320, 12, dev)
a, b = synth.phantom_pair(8, 1, 320, 320, seed=1234); a, b = a.to(dev), b.to(dev)
net.train()
for _ in range(3): bench.train_step(net, a, b)
cnt = collections.Counter(); reg = collections.Counter()
orig_one = ops.PACKS._pack_one; orig_reg = ops.PACKS._register
def one(job, w):
    cnt[(tuple(w.shape), job["mode"], job["version"], w._version)] += 1
    return orig_one(job, w)
def regf(w, mode):
    reg[(tuple(w.shape), mode, type(w).__name__)] += 1
    return orig_reg(w, mode)
ops.PACKS._pack_one = one; ops.PACKS._register = regf
bench.train_step(net, a, b)
torch.cuda.synchronize()
print("pack_one calls:", sum(cnt.values()), "registrations:", sum(reg.values()), "jobs:", len(ops.PACKS.jobs))
for k, v in list(cnt.items())[:12]: print("  one", k, v)
for k, v in list(reg.items())[:12]: print("  reg", k, v)
