import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for cin, cout, s in ((36, 36, 160), (18, 18, 320), (36, 18, 320)):
    N = 8
    x = torch.randn(N, cin, s, s, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    sc = torch.rand(N, cin, device=dev) + 0.5; sh = torch.randn(N, cin, device=dev)
    y = torch.empty(N, cout, s, s, device=dev)
    part = ops.conv2d(ops.Act(x, 0, cin, sc, sh, 0.2), wt, None, ops.full(y), stats=True)
    torch.cuda.synchronize()
    cnt, mean, m2 = part[..., 0].double(), part[..., 1].double(), part[..., 2].double()
    tot = cnt.sum(-1); mu = (cnt * mean).sum(-1) / tot
    yd = y.double(); tm = yd.mean((2, 3))
    err = (mu - tm).abs()
    print(cin, cout, s, "tot ok", bool((tot == s * s).all()), "max mean err", err.max().item(), "planes off >1e-6:", int((err > 1e-6).sum()), "of", err.numel())
    bad = (err > 1e-6).nonzero()[:6]
    print("  bad planes (n, c):", bad.tolist())
    nz = (cnt > 0).sum(-1)
    print("  records with count>0 per plane: min", int(nz.min()), "max", int(nz.max()), "; counts seen:", sorted(set(cnt.flatten().tolist()))[:8])
