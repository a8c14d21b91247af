"""What is left in a recorded training step besides C-ABI calls: the Python callables (torch operations) by name, and the tape segments."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
import bench
from spatialalignmentnetwork_amd import synth
dev = torch.device("cuda:0")
net = bench.build_model(8, 320, 320, 12, dev).train()
a, b = (t.to(dev) for t in synth.phantom_pair(8, 1, 320, 320, seed=1234))
step = net.record_update(a, b, warmup=2, restore=True)
kinds = collections.Counter()
names = collections.Counter()
for fn, args, kind in step.calls:
    if kind:
        kinds["C-ABI"] += 1
        continue
    if isinstance(fn, ctypes._CFuncPtr):
        kinds["C-ABI (no rc)"] += 1
        continue
    owner = getattr(fn, "__self__", None)
    nm = getattr(fn, "__name__", str(fn))
    label = f"{type(owner).__name__}.{nm}" if owner is not None else nm
    kinds["python"] += 1
    names[label] += 1
print(dict(kinds))
for k, v in names.most_common(40):
    print(f"{v:5d}  {k}")
segs = step._compile()
print("segments:", len(segs), "tapes:", sum(1 for s in segs if not isinstance(s, tuple)), "python callables between them:", sum(1 for s in segs if isinstance(s, tuple)))
for s in segs:
    if isinstance(s, tuple):
        fn = s[0]; owner = getattr(fn, "__self__", None)
        print("   py:", type(owner).__name__ if owner is not None else "", getattr(fn, "__name__", fn), [tuple(x.shape) if hasattr(x, "shape") else x for x in s[1]][:3])
