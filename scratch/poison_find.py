"""SAN_ARENA_POISON=1: which tensor turns NaN first in the configuration of test_update_data_parallel_two_ranks (one process)?"""
import os, sys, torch
os.environ["SAN_ARENA_POISON"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import basemodel, model as smodel, synth, ops
DEV = "cuda:0"
reg = sys.argv[1] if len(sys.argv) > 1 else "None"
h, w = 48, 80
cfg = basemodel.Config(sparsity=0.25, lr=1e-4, shape=w, coils=1, reg=reg, mask="equispaced", weight_smooth=1000.0, weight_gan=0.0,
                       weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=4, sens_chans=2, pools=2, sens_pools=2)
net = smodel.CSModel(cfg); net.net_mask.pruned = synth.equispaced_pruned(w, 0.25, 0)
for sub, sd in (("net_T", 41), ("net_R", 42)):
    m = getattr(net, sub); m.load_state_dict(synth.fill_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=sd))
net.to(DEV).train(); net.net_T.eval(); net.auto_record = False
xf, xa = (t.to(DEV).contiguous() for t in synth.phantom_pair(1, 1, h, w, seed=40))
def nan(t): return bool(torch.isnan(t).any())
for step in range(2):
    net.set_input(xf, xa)
    print("step", step, "inputs:", {k: nan(getattr(net, k)) for k in ("img_k_sampled", "img_sampled")})
    net.update(); torch.cuda.synchronize()
    print("  outputs:", {k: nan(getattr(net, k)) for k in ("img_offset", "img_warped", "img_rec") if hasattr(net, k)}, "loss_sim", float(net.loss_sim))
    bad = [n_ for n_, p in net.net_R.named_parameters() if p.grad is not None and nan(p.grad)]
    print("  NaN grads in net_R:", len(bad), bad[:6])
    badp = [n_ for n_, p in net.net_R.named_parameters() if nan(p)]
    print("  NaN params in net_R:", len(badp), badp[:4])
