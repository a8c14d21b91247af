import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spatialalignmentnetwork_amd import ops
dev = "cuda:0"
def bench(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, c, h, w in [(8, 18, 320, 320), (8, 36, 160, 160), (8, 18, 160, 160)]:
    g = torch.randn(n, c, h, w, device=dev); y = torch.randn(n, c, h, w, device=dev)
    sc = torch.rand(n, c, device=dev) + 0.5; sh = torch.randn(n, c, device=dev)
    dy = torch.empty_like(g)
    g2 = torch.randn(n, c, h // 2, w // 2, device=dev)
    ops.AMAX.reset(dev)
    t = bench(lambda: ops.act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True))
    t2 = bench(lambda: ops.act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True, g2=ops.full(g2)))
    print(f"act_bwd {n}x{c}x{h}x{w}: {t:.1f} us, with second source {t2:.1f} us", flush=True)
