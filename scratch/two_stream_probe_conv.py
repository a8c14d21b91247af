"""A data-gradient convolution (conv_bf16x3_kernel: 132 packed-fp32 instructions with op_sel) on one stream while OTHER
convolutions run on another stream, nothing in common: launches whose output differs from the launch that ran alone.
usage: PYTHONPATH=. python scratch/two_stream_probe_conv.py"""
import torch
from spatialalignmentnetwork_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
aux = torch.cuda.Stream()
ar_v, ar_a = ops.Arena(), ops.Arena()
victims = [(torch.randn(co, ci, 3, 3, device=dev) * 0.05, torch.randn(1, co, hh, ww, device=dev), torch.empty(1, ci, hh, ww, device=dev))
           for co, ci, hh, ww in ((32, 16, 640, 368), (64, 32, 320, 184), (128, 64, 160, 92), (256, 128, 80, 46))]
wgt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
dy, dx = torch.randn(1, 64, 160, 92, device=dev), torch.empty(1, 64, 160, 92, device=dev)
for w_, dy_, dx_ in victims:
    with ops.use_arena(ar_v):
        ops.conv2d_dgrad(ops.full(dy_), w_, ops.full(dx_))
    torch.cuda.synchronize()
    want = dx_.clone()
    for beside in (False, True):
        bad, worst = 0, 0.0
        for it in range(150):
            if beside:
                with ops.use_arena(ar_a):
                    for _ in range(4):
                        ops.conv2d_dgrad(ops.full(dy), wgt, ops.full(dx))
            with torch.cuda.stream(aux), ops.use_arena(ar_v):
                ops.conv2d_dgrad(ops.full(dy_), w_, ops.full(dx_))
            torch.cuda.synchronize()
            if not torch.equal(dx_, want):
                bad += 1
                worst = max(worst, float((dx_ - want).abs().max() / want.abs().max()))
        print(f"victim {tuple(w_.shape[:2])} @ {dy_.shape[2]}x{dy_.shape[3]}, convolutions beside = {beside}: {bad} of 150 launches differ (worst {worst:.1e})", flush=True)
