"""Which weights are re-packed one by one (not by the batched launch) in a steady-state training step?"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spatialalignmentnetwork_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
net = bench.build_model(2, 320, 320, 12, dev)
net.train()
full, aux = (t.to(dev) for t in synth.phantom_pair(2, 1, 320, 320, seed=1234))
names = {}
for sub in ("net_T", "net_R"):
    for k, p in getattr(net, sub).named_parameters():
        names[p.data_ptr()] = f"{sub}.{k}"
hits = collections.Counter()
for reg in (ops.PACKS, ops.PACKS16):
    orig = reg._pack_one

    def wrap(job, w, orig=orig, reg=reg):
        hits[(type(reg).__name__, names.get(w.data_ptr(), "?"), job["mode"], tuple(w.shape))] += 1
        return orig(job, w)
    reg._pack_one = wrap
for step in range(4):
    if step == 2:
        hits.clear()
    bench.train_step(net, full, aux)
torch.cuda.synchronize()
print(len(hits), "distinct single packs over 2 steady-state steps")
for k, v in sorted(hits.items(), key=lambda kv: str(kv[0]))[:60]:
    print(v, k)
