"""The data-parallel step under a ONE-rank RCCL group (SAN_DIST_SINGLE=1), stage by stage, with faulthandler."""
import faulthandler, os, sys, types
faulthandler.enable()
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", SAN_DIST_SINGLE="1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import basemodel, dist as sdist, model as smodel, synth
dev = torch.device("cuda:0")
d = sdist.init("nccl", dev)
print("init ok, backend", sdist.backend(), flush=True)
w = 80
cfg = basemodel.Config(sparsity=0.25, lr=1e-4, shape=w, coils=1, reg="Rec", mask="equispaced", weight_smooth=1000.0, weight_gan=0.0,
                       weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=2, chans=4, sens_chans=8, pools=2, sens_pools=2)
net = smodel.CSModel(cfg)
net.net_mask.pruned = synth.equispaced_pruned(w, 0.25, 0)
net.to(dev).train()
net.time_exchange = True
stages = int(os.environ.get("STAGES", "5"))
for it in range(stages):
    net.set_input(*(t.to(dev).contiguous() for t in synth.phantom_pair(2, 1, 48, w, seed=300 + it)))
    net.update()
    torch.cuda.synchronize()
    print("step", it, net.step_mode, getattr(net, "exchange_slices", None), flush=True)
if os.environ.get("CAPTURE", "1") == "1":
    xf, xa = (t.to(dev).contiguous() for t in synth.phantom_pair(2, 1, 48, w, seed=310))
    cap = net.capture_update(xf, xa, warmup=1)
    cap.replay(); torch.cuda.synchronize()
    print("capture", cap.mode, flush=True)
d.destroy_process_group()
print("done", flush=True)
