"""One backward pass at the multi-coil shape (or the bench shape): flat gradients of both networks with every bf16x3
kernel on vs. the fp32 kernels, norm-wise; and the forward outputs."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import synth, ops
from spatialalignmentnetwork_amd.basemodel import Config
from spatialalignmentnetwork_amd.model import CSModel
dev = torch.device('cuda', 0)
n, c, h, w = (1, 15, 640, 368) if "--mc" in sys.argv else (8, 1, 320, 320)
sp = 0.125 if c > 1 else 0.25
cfg = Config(sparsity=sp, lr=1e-4, shape=w, coils=c, reg="Rec", mask="equispaced", weight_smooth=1000.0,
             weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=12)
net = CSModel(cfg)
net.net_mask.pruned = synth.equispaced_pruned(w, sp, 0)
for sub, sd in (("net_T", 1), ("net_R", 2)):
    m = getattr(net, sub)
    m.load_state_dict(synth.fill_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=sd))
net.to(dev).train()
a, b = synth.phantom_pair(n, c, h, w, seed=7)
a, b = a.to(dev), b.to(dev)
res = {}
torch.manual_seed(0)
noise = 1.0 + 3e-7 * torch.randn(a.shape, device=dev)
for name, flag in (("bf16x3", True), ("fp32", False), ("fp32p", False), ("bf16x3p", True)):
    ops.USE_BF16X3[0] = flag
    ops.bump_weight_epoch()
    net.set_input(a * noise if name.endswith('p') else a, b); net.loss_all = 0
    net.forwardT(); net.forwardR()
    for o in (net.optim_R, net.optim_T): o.zero_grad()
    with ops.wgrad_overlap():
        net.backward(True)
    torch.cuda.synchronize()
    res[name] = {"rec": net.img_rec.clone(), "loss": net.loss_sim.item()}
    res[name]["gR"] = torch.cat([p.grad.reshape(-1) for p in net.net_R.parameters()]).clone()
    res[name]["gT"] = torch.cat([p.grad.reshape(-1) for p in net.net_T.parameters()]).clone()
def rel(x, y): return ((x.double() - y.double()).norm() / y.double().norm()).item()
print("rec", rel(res["bf16x3"]["rec"], res["fp32"]["rec"]), "loss", res["bf16x3"]["loss"], res["fp32"]["loss"])
print("grad R", rel(res["bf16x3"]["gR"], res["fp32"]["gR"]), " grad T", rel(res["bf16x3"]["gT"], res["fp32"]["gT"]))
print("fp32 vs fp32 with 3e-7 input noise:   rec", rel(res["fp32p"]["rec"], res["fp32"]["rec"]), " grad R", rel(res["fp32p"]["gR"], res["fp32"]["gR"]), " grad T", rel(res["fp32p"]["gT"], res["fp32"]["gT"]))
print("bf16x3 vs bf16x3 with 3e-7 input noise: rec", rel(res["bf16x3p"]["rec"], res["bf16x3"]["rec"]), " grad R", rel(res["bf16x3p"]["gR"], res["bf16x3"]["gR"]), " grad T", rel(res["bf16x3p"]["gT"], res["bf16x3"]["gT"]))
