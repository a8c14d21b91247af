"""Which path deviates with the sensitivity-network overlap: eager or replay?  (states after k steps vs the overlap-off eager run)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import basemodel, model as smodel, synth, ops
DEV = "cuda:0"
n, c, h, w = 2, 3, 48, 80
def make():
    cfg = basemodel.Config(sparsity=0.25, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0, weight_gan=0.0,
                           weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=18, sens_chans=8, pools=2, sens_pools=2)
    net = smodel.CSModel(cfg)
    net.net_mask.pruned = synth.equispaced_pruned(w, 0.25, 0)
    for sub, sd in (("net_T", 41), ("net_R", 42)):
        m = getattr(net, sub)
        m.load_state_dict(synth.fill_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=sd))
    net.to(DEV).train()
    net.auto_record = False
    for o in (net.optim_R, net.optim_T):
        o.device_step = True
    return net
def state(net):
    return {f"{s_}.{k}": v.detach().cpu().clone() for s_ in ("net_R", "net_T") for k, v in getattr(net, s_).state_dict().items()}
def grads(net):
    return {nm: p.grad.detach().cpu().clone() for nm, p in net.net_R.named_parameters() if p.grad is not None}

if os.environ.get("PREAMBLE", "1") == "1":
    # what tests/test_hip_parity_r3.py::test_amax_pool_resets_on_an_unindexed_device leaves behind
    ops.AMAX.reset(torch.device("cuda"))
    rec_ = ops.AMAX.next(torch.device("cuda")); rec_ is not None and rec_.fill_(0x7F000000); ops.AMAX.reset(torch.device("cuda"))
    rec_ = ops.AMAX.next(DEV); rec_ is not None and rec_.fill_(0x7F000000); ops.AMAX.reset("cuda")
    f0, a0 = synth.phantom_pair(2, 1, 32, 32, seed=40)
    for dev in (torch.device("cuda"), torch.device(DEV)):
        cfg0 = basemodel.Config(sparsity=0.25, lr=1e-4, shape=32, coils=1, reg="Rec", mask="equispaced", weight_smooth=1000.0, weight_gan=0.0,
                                weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=4, sens_chans=2, pools=2, sens_pools=2)
        net0 = smodel.CSModel(cfg0); net0.net_mask.pruned = synth.equispaced_pruned(32, 0.25, 0)
        net0 = net0.to(dev).train()
        for _ in range(2):
            net0.set_input(f0.to(dev), a0.to(dev)); net0.update()
        torch.cuda.synchronize()
    del net0
xf, xa = (t.to(DEV).contiguous() for t in synth.phantom_pair(n, c, h, w, seed=40))
def cmp(a, b, label):
    bad = [(k, (a[k].double() - b[k].double()).abs().max().item()) for k in a if not torch.equal(a[k], b[k])]
    print(f"{label}: {len(bad)} of {len(a)} differ", [(k[-60:], f"{e:.2e}") for k, e in bad[:4]], flush=True)
res = {}
for ov in (False, True):
    smodel.SENS_OVERLAP[0] = ov
    net = make()
    net.set_input(xf, xa); net.update(); torch.cuda.synchronize()
    res[("eager", ov, "g1")] = grads(net)
    res[("eager", ov, 1)] = state(net)
    for _ in range(2):
        net.set_input(xf, xa); net.update()
    torch.cuda.synchronize()
    res[("eager", ov, 3)] = state(net)
    cap = make()
    step = cap.record_update(xf, xa, warmup=2)
    step.replay(); torch.cuda.synchronize()
    res[("replay", ov, 1)] = state(cap)
    res[("replay", ov, "g1")] = grads(cap)
    step.replay(); step.replay(); torch.cuda.synchronize()
    res[("replay", ov, 3)] = state(cap)
base = res[("eager", False, 1)]
cmp(res[("eager", False, "g1")], res[("eager", True, "g1")], "grads after step 1: eager overlap vs eager plain")
cmp(res[("eager", False, "g1")], res[("replay", False, "g1")], "grads after step 1: replay plain vs eager plain")
cmp(res[("eager", False, "g1")], res[("replay", True, "g1")], "grads after step 1: replay overlap vs eager plain")
for k in (1, 3):
    cmp(res[("eager", False, k)], res[("eager", True, k)], f"state after {k}: eager overlap vs eager plain")
    cmp(res[("eager", False, k)], res[("replay", False, k)], f"state after {k}: replay plain vs eager plain")
    cmp(res[("eager", False, k)], res[("replay", True, k)], f"state after {k}: replay overlap vs eager plain")
print("---- auto-record scenario", flush=True)
for ov in (False, True):
    smodel.SENS_OVERLAP[0] = ov
    net = make(); net.auto_record = True
    modes = []
    for _ in range(3):
        net.set_input(xf, xa); net.update(); modes.append(net.step_mode[:6])
    torch.cuda.synchronize()
    cmp(res[("eager", False, 3)], state(net), f"auto-record (overlap {ov}) {modes} after 3 vs eager plain")
    net = make(); net.auto_record = True
    for i in range(2):
        net.set_input(xf, xa); net.update()
    torch.cuda.synchronize()
    s2 = state(net)
    st = net._auto_state()
    net._auto_busy = True
    try:
        net._auto_make(st)
    finally:
        net._auto_busy = False
    torch.cuda.synchronize()
    cmp(s2, state(net), f"   (overlap {ov}) the recording itself changed the state")
