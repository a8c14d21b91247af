"""san_dc_rows (image-domain cascade boundary) at a given shape: us per call and GB/s on the kernel's own bytes.
    python scratch/bench_dc_rows.py [N C H W ...]      (default: the multi-coil 640 x 368 x 15 case and the 320 x 320 case)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops  # noqa: E402

dev = "cuda:0"


def bench(fn, reps=30):
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


shapes = [(1, 15, 640, 368), (2, 15, 640, 368), (8, 1, 320, 320), (8, 1, 640, 368), (1, 15, 320, 320)]
if len(sys.argv) > 4:
    shapes = [tuple(int(v) for v in sys.argv[1:5])]
for N, C, H, W in shapes:
    x = torch.randn(N, C, H, W, dtype=torch.complex64, device=dev)
    s, k0 = torch.randn_like(x), torch.randn_like(x)
    mask = (torch.rand(W, device=dev) > 0.5).float()
    dcw = torch.ones(1, device=dev)
    r = torch.randn(N, 2, H, W, device=dev)
    xo, m = torch.empty_like(x), torch.empty(N, 3, H, W, device=dev)
    t = bench(lambda: ops.dc_rows(x, s, k0, mask, dcw, r, xo, m))
    own = (4 * C + 2) * N * H * W * 8
    print(f"dc_rows N={N} C={C} {H}x{W}: {t:7.1f} us  {own / t / 1e3:6.0f} GB/s on its own (4C+2) planes, "
          f"{(6 * C + 2) * N * H * W * 8 / t / 1e3:6.0f} GB/s on the SURVEY 8(d) count", flush=True)
    if os.environ.get("DC_TRAIN"):
        dk = torch.empty_like(x)
        t1 = bench(lambda: ops.dc_rows(x, s, k0, mask, dcw, r, xo, m, dk_out=dk))
        gw = torch.zeros(1, device=dev)
        t2 = bench(lambda: ops.dc_rows_bwd(x, s, mask, dcw, xo, m, dk, gw))
        print(f"   training forward (+ dk_out) {t1:7.1f} us   backward {t2:7.1f} us", flush=True)
