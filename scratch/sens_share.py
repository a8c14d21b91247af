"""Pointers that calls on the auxiliary stream share with calls on the other streams inside the windows where they run concurrently."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import basemodel, model as smodel, synth, ops, _lib
DEV = "cuda:0"
n, c, h, w = 2, 3, 48, 80
cfg = basemodel.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0, weight_gan=0.0,
                       weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=18, sens_chans=8, pools=2, sens_pools=2)
net = smodel.CSModel(cfg)
net.net_mask.pruned = synth.equispaced_pruned(w, 0.25, 0)
net.to(DEV).train()
xf, xa = (t.to(DEV).contiguous() for t in synth.phantom_pair(n, c, h, w, seed=40))
step = net.record_update(xf, xa, warmup=2)
aux = net._aux_stream.cuda_stream
protos = _lib.lib().protos
# windows: from an "aux.wait_stream(main)" to the next "main.wait_stream(aux)"
events = []
for i, (fn, args, kind) in enumerate(step.calls):
    name = getattr(fn, "__name__", "?")
    owner = getattr(fn, "__self__", None)
    if isinstance(owner, torch.cuda.Stream) and name == "wait_stream":
        events.append((i, "fork" if owner.cuda_stream == aux else ("join" if args[0].cuda_stream == aux else "other")))
print([e for e in events if e[1] != "other"])
forks = [i for i, k in events if k == "fork"]; joins = [i for i, k in events if k == "join"]
def ptrs(i):
    fn, args, kind = step.calls[i]
    name = getattr(fn, "__name__", "?")
    if name not in protos: return None, None, []
    at = protos[name][1]
    vals = []
    for a, t in zip(args, at):
        if t is ctypes.c_void_p:
            v = a.value if isinstance(a, ctypes.c_void_p) else a
            if v: vals.append(int(v))
    stream = vals[-1] if vals else None
    return name, stream, vals[:-1]
for f in forks:
    j = min([x for x in joins if x > f], default=len(step.calls))
    auxp, othp = {}, {}
    for i in range(f, j):
        name, st, ps = ptrs(i)
        if name is None: continue
        d = auxp if st == aux else othp
        for p in ps: d.setdefault(p, set()).add(name)
    shared = set(auxp) & set(othp)
    print(f"window calls {f}..{j}: aux ptrs {len(auxp)}, other ptrs {len(othp)}, shared {len(shared)}")
    for p in list(shared)[:12]:
        print("   ", hex(p), sorted(auxp[p])[:3], "|", sorted(othp[p])[:3])
print("---- the shared pointer's argument positions")
for f in forks[:1]:
    j = min([x for x in joins if x > f], default=len(step.calls))
    for i in range(f, j):
        fn, args, kind = step.calls[i]
        name = getattr(fn, "__name__", "?")
        if name in ("san_fft2", "san_fft_cols"):
            vals = [(a.value if isinstance(a, ctypes.c_void_p) else a) for a in args]
            print(i, name, ["aux" if vals[-1] == aux else "main"], [hex(v) if isinstance(v, int) and v and v > 1 << 32 else v for v in vals])
print("img_k_sampled", hex(net.img_k_sampled.data_ptr()))
print("---- storages shared between the auxiliary stream and the other streams in the backward window")
import gc
stor = {}
for o in gc.get_objects():
    try:
        if isinstance(o, torch.Tensor) and o.is_cuda:
            st_ = o.untyped_storage()
            stor[st_.data_ptr()] = (st_.nbytes(), tuple(o.shape), str(o.dtype))
    except Exception:
        pass
import bisect
bases = sorted(stor)
def owner_of(p):
    i = bisect.bisect_right(bases, p) - 1
    if i >= 0 and p < bases[i] + stor[bases[i]][0]:
        return bases[i]
    return None
f = forks[-1]; j = min([x for x in joins if x > f], default=len(step.calls))
auxs, oths = {}, {}
for i in range(f, j):
    name, st, ps = ptrs(i)
    if name is None: continue
    d = auxs if st == aux else oths
    for p in ps:
        b = owner_of(p)
        d.setdefault(b, []).append((name, p - b if b else p))
for b in set(auxs) & set(oths):
    if b is None:
        print("unknown storage pointers on both sides:", len(auxs[b]), len(oths[b])); continue
    a_offs = sorted({o for _, o in auxs[b]}); o_offs = sorted({o for _, o in oths[b]})
    print(hex(b), stor[b], "aux:", sorted({n for n, _ in auxs[b]})[:4], a_offs[:6], "| other:", sorted({n for n, _ in oths[b]})[:4], o_offs[:6], "..." if len(o_offs) > 6 else "")
