"""Round 6: a convolution + its InstanceNorm finalisation, in-kernel (last arriver) vs as a second launch, per layer (us per pair)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spatialalignmentnetwork_amd import ops
dev = "cuda:0"
torch.manual_seed(0)


def bench(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


LAYERS = [(8, 18, 18, 320, 320, 3), (8, 36, 18, 320, 320, 3), (8, 36, 36, 160, 160, 3), (8, 72, 36, 160, 160, 3), (8, 72, 72, 80, 80, 3),
          (8, 144, 72, 80, 80, 3), (8, 144, 144, 40, 40, 3), (8, 288, 144, 40, 40, 3), (8, 288, 288, 20, 20, 3), ("t", 8, 288, 144, 20, 20),
          ("t", 8, 72, 36, 80, 80), ("t", 8, 36, 18, 160, 160)]
for L in LAYERS:
    row = []
    for on in (False, True):
        ops.FIN_INKERNEL[0] = on
        if L[0] == "t":
            _, n, cin, cout, h, w = L
            x = torch.randn(n, cin, h, w, device=dev)
            wt = torch.randn(cin, cout, 2, 2, device=dev) * 0.1
            y = ops.Act(torch.empty(n, cout, 2 * h, 2 * w, device=dev), 0, cout, torch.empty(n, cout, device=dev), torch.empty(n, cout, device=dev), 0.2)

            def fn():
                part = ops.tconv2x2(ops.full(x), wt, y, stats=True, tag=".fl", instance_norm_eps=1e-5)
                if part is not None:
                    ops.norm_finalize(part, ops.NORM_INSTANCE, 1e-5, y.scale, y.shift, y.coff)
        else:
            n, cin, cout, h, w, ks = L
            x = torch.randn(n, cin, h, w, device=dev)
            wt = torch.randn(cout, cin, ks, ks, device=dev) * 0.05
            y = ops.Act(torch.empty(n, cout, h, w, device=dev), 0, cout, torch.empty(n, cout, device=dev), torch.empty(n, cout, device=dev), 0.2)

            def fn():
                part = ops.conv2d(ops.full(x), wt, None, y, stats=True, instance_norm_eps=1e-5, tag=".fl")
                if part is not None:
                    ops.norm_finalize(part, ops.NORM_INSTANCE, 1e-5, y.scale, y.shift, y.coff)
        row.append(bench(fn))
    print(f"{str(L):34s} two launches {row[0]:6.1f} us   in-kernel {row[1]:6.1f} us", flush=True)
