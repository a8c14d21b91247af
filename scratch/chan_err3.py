import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
cin, cout, s, N = 18, 18, 160, 1
def run(x, wt):
    y = torch.empty(N, cout, s, s, device=dev)
    ops.conv2d(ops.full(x), wt, None, ops.full(y), stats=False); torch.cuda.synchronize()
    want = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1)
    e = ((y.double() - want) ** 2).sum((0, 2, 3)).sqrt() / (want ** 2).sum((0, 2, 3)).sqrt()
    return " ".join(f"{v:.1e}" for v in e.tolist()[13:])
x = torch.randn(N, cin, s, s, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
print("x full, w fp16-exact :", run(x, wt.half().float()))
print("x fp16-exact, w full :", run(x.half().float(), wt))
print("x full, w full       :", run(x, wt))
xs = x.half().float(); xl = torch.zeros_like(x); xl[:, 5] = 2.0 ** -13
print("x lo on channel 5 only, w exact:", run(xs + xl, wt.half().float()))
