"""Per-layer timings of the bf16x3 convolution over the layer mix of one cascade U-Net + the alignment net (N = 8):
    python scratch/bench_layers.py [conv|wgrad]
Prints one line per layer (us, algorithmic TFLOP/s) and the mix total.  Used with scratch/ab_lib.sh for same-box A/B."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops  # noqa: E402

dev = "cuda:0"
N = int(os.environ.get("BL_N", "8"))
LAYERS = [  # (cin, cout, size, count per cascade pass)
    (18, 18, 320, 2), (36, 18, 320, 1), (18, 36, 160, 1), (36, 36, 160, 2), (72, 36, 160, 1), (36, 72, 80, 1), (72, 72, 80, 2),
    (144, 72, 80, 1), (72, 144, 40, 1), (144, 144, 40, 2), (288, 144, 40, 1), (144, 288, 20, 1), (288, 288, 20, 1),
    (32, 32, 320, 0.25), (96, 32, 320, 0.08), (64, 64, 160, 0.5), (128, 64, 160, 0.08)]


if os.environ.get("BL_ONLY"):                      # e.g. BL_ONLY=128-64-160,64-64-160
    LAYERS = [tuple(int(v) for v in t.split("-")) + (1,) for t in os.environ["BL_ONLY"].split(",")]


def bench(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


which = sys.argv[1] if len(sys.argv) > 1 else "conv"
if len(sys.argv) > 2:
    ops.set_conv_precision(sys.argv[2])          # bf16x3 | bf16x2 | bf16
STATS = os.environ.get("BL_STATS", "1") == "1"
total = 0.0
for cin, cout, s, cnt in LAYERS:
    x = torch.randn(N, cin, s, s, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    sc = torch.rand(N, cin, device=dev) + 0.5
    sh = torch.randn(N, cin, device=dev)
    y = torch.empty(N, cout, s, s, device=dev)
    xa = ops.Act(x, 0, cin, sc, sh, 0.2)
    if which == "conv":
        t = bench(lambda: ops.conv2d(xa, wt, None, ops.full(y), stats=STATS))
    else:
        dw = torch.zeros_like(wt)
        dy = torch.randn(N, cout, s, s, device=dev)
        dya = ops.full(dy)
        if ops.F16_BWD[0]:                        # the two-fp16-part form needs the recorded max |dy| (bits of the float)
            dya.amax = ops.amax_record(dy.abs().max())
        t = bench(lambda: ops.conv2d_wgrad(xa, dya, dw, accumulate=True))
    fl = 2.0 * N * s * s * cin * cout * 9
    total += t * cnt
    print(f"{which} {cin:3d}->{cout:3d} @{s:3d}: {t:8.1f} us {fl / t / 1e6:6.1f} TF", flush=True)
print(f"{which} mix (one cascade pass + alignment share)  : {total:8.1f} us", flush=True)
