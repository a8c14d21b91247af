"""Per-wave phase cycle counts of the v3 stream kernel (library built with SAN_EXTRA_HIPCC_FLAGS=-DSAN_STREAM_DBG).
    SAN_CONV_STREAM=3 python scratch/stream_dbg.py 18-18-320"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
from spatialalignmentnetwork_amd import _lib
dev = "cuda:0"
cin, cout, s = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "18-18-320").split("-"))
N = 8
torch.manual_seed(0)
x = torch.randn(N, cin, s, s, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
sc = torch.rand(N, cin, device=dev) + 0.5; sh = torch.randn(N, cin, device=dev)
y = torch.empty(N, cout, s, s, device=dev)
xa = ops.Act(x, 0, cin, sc, sh, 0.2)
for _ in range(3):
    ops.conv2d(xa, wt, None, ops.full(y), stats=True)
torch.cuda.synchronize()
cdll = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libsan_hip.so"))
buf = np.zeros(512 * 4 * 10, dtype=np.uint64)
rc = cdll.san_conv_stream_dbg_dump(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
d = buf.reshape(-1, 10).astype(np.float64)
d = d[d[:, 8] > 0]
names = ["prologue", "wait barrier 1", "stage", "wait barrier 2 (+w dma)", "next item + prefetch issue", "K-loop", "prev copy", "final epilogue"]
print(f"rc {rc}; waves {len(d)}; wave lifetime mean {d[:, 8].mean():.0f} cycles (min {d[:, 8].min():.0f} max {d[:, 8].max():.0f})")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {d[:, i].mean():9.0f} cycles  {100 * d[:, i].mean() / d[:, 8].mean():5.1f} %")
