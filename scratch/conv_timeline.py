"""Workgroup timeline of ONE bf16x3 / fp16-part convolution launch (san_conv_bf16x3_debug_timeline):
    python scratch/conv_timeline.py cin cout size [N]
Every workgroup records the 100 MHz clock at start / first chunk staged / epilogue start / end and its HW_ID, XCC_ID.
Prints the launch span, the phase times of early and late workgroups and how many workgroups each CU ran at once."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops  # noqa: E402
from spatialalignmentnetwork_amd._lib import lib  # noqa: E402

cin, cout, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = "cuda:0"
x = torch.randn(N, cin, s, s, device=dev)
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
sc, sh = torch.rand(N, cin, device=dev) + 0.5, torch.randn(N, cin, device=dev)
y = torch.empty(N, cout, s, s, device=dev)
xa = ops.Act(x, 0, cin, sc, sh, 0.2)
for _ in range(3):
    ops.conv2d(xa, wt, None, ops.full(y), stats=True)
torch.cuda.synchronize()
buf = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
lib().call("san_conv_bf16x3_debug_timeline", buf.data_ptr())
ops.conv2d(xa, wt, None, ops.full(y), stats=True)
torch.cuda.synchronize()
lib().call("san_conv_bf16x3_debug_timeline", None)
b = buf.cpu().numpy().reshape(-1, 8)
b = b[b[:, 0] != 0]
t = (b[:, :4] - b[:, 0].min()) * 0.01                      # us
if b[:, 6].any():
    k0, k3 = (b[:, 6] - b[:, 1]) * 0.01, (b[:, 7] - b[:, 6]) * 0.01
    print(f"  first chunk: barrier -> K-step 0 issued {k0.mean():.2f} us, K-steps 1..3 {k3.mean():.2f} us")
hw, xcc = b[:, 4], b[:, 5] & 15
cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)
print(f"{cin}->{cout} @{s} N={N}: {len(b)} workgroups on {len(np.unique(cu))} CUs, span {t[:, 3].max():.1f} us")
order = np.argsort(t[:, 0])
t, cu = t[order], cu[order]
d = np.diff(t, axis=1)
for name, sel in (("first 256", slice(0, 256)), ("256..768", slice(256, 768)), ("768..", slice(768, None))):
    if len(t[sel]) == 0:
        continue
    print(f"  {name:10s} n={len(t[sel]):5d} start {t[sel, 0].mean():6.1f} (max {t[sel, 0].max():6.1f})  load+stage {d[sel, 0].mean():5.1f}  "
          f"k-loop {d[sel, 1].mean():5.1f}  epilogue {d[sel, 2].mean():5.1f}  total {(t[sel, 3] - t[sel, 0]).mean():5.1f} us")
# concurrency per CU over time
ev = sorted([(a, 1, c) for a, c in zip(t[:, 0], cu)] + [(e, -1, c) for e, c in zip(t[:, 3], cu)])
tot, last, area = 0, 0.0, {}
for when, dlt, _ in ev:
    area[tot] = area.get(tot, 0.0) + (when - last)
    tot, last = tot + dlt, when
span = t[:, 3].max()
print("  resident workgroups (chip-wide) : share of the span   " + "  ".join(f"{k // 64 * 64:4d}+:{sum(v for kk, v in area.items() if kk // 64 == k // 64) / span:5.2f}" for k in sorted(set(kk // 64 * 64 for kk in area))))
per = np.bincount(cu.astype(np.int64))
print("  workgroups per CU: min", per[per > 0].min(), "max", per.max(), " end-time quantiles (us):", np.round(np.quantile(t[:, 3], [0.1, 0.5, 0.9, 0.99, 1.0]), 1))
