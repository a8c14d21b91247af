"""act_bwd (InstanceNorm backward of whole planes) on one stream while data-gradient convolutions run on another: no model, no
memory in common.  Counts the act_bwd launches whose output differs from the same launch run alone.
usage: python scratch/two_stream_probe.py [fused|unfused] [wd]      (wd: convolution weights direct from memory -- 50 KB of LDS)"""
import os
import sys
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
if mode == "unfused":
    os.environ["SAN_NO_ACT_BWD_FUSED"] = "1"
import torch
from spatialalignmentnetwork_amd import ops
from spatialalignmentnetwork_amd.ops import Act

dev = torch.device("cuda:0")
torch.manual_seed(0)
if "wd" in sys.argv:
    ops.lib().call("san_conv_bf16x3_set_tuning", 1, -1)
main, aux = torch.cuda.current_stream(), torch.cuda.Stream()
# victim: the sensitivity net's 160 x 92 level, 15 coils x 32 channels
n, c, h, w = 15, 32, 160, 92
gbuf, y = torch.randn(n, c, h, w, device=dev), torch.randn(n, c, h, w, device=dev)
sc, sh = torch.rand(n, c, device=dev) + 0.5, torch.randn(n, c, device=dev) * 0.1
out = torch.empty_like(gbuf)
ar_v, ar_a = ops.Arena(), ops.Arena()
# aggressor: 64 -> 64 channels at 160 x 92 (the dummy set 0 of the model-level experiment)
wgt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
dy, dx = torch.randn(1, 64, 160, 92, device=dev), torch.empty(1, 64, 160, 92, device=dev)


def victim():
    ops.act_bwd(ops.full(gbuf), Act(y, 0, c, sc, sh, 0.2), ops.full(out), instance_norm=True)


with ops.use_arena(ar_v):
    victim()
torch.cuda.synchronize()
want = out.clone()
with ops.use_arena(ar_a):
    ops.conv2d_dgrad(ops.full(dy), wgt, ops.full(dx))
torch.cuda.synchronize()
for beside in (False, True):
    bad = 0
    worst = 0.0
    for it in range(200):
        if beside:
            with ops.use_arena(ar_a):
                for _ in range(4):
                    ops.conv2d_dgrad(ops.full(dy), wgt, ops.full(dx))
        with torch.cuda.stream(aux), ops.use_arena(ar_v):
            victim()
        torch.cuda.synchronize()
        if not torch.equal(out, want):
            bad += 1
            worst = max(worst, float((out - want).abs().max() / want.abs().max()))
            if bad <= 3:
                d = (out - want).reshape(n * c, -1)
                planes = (d != 0).any(1).nonzero().flatten().tolist()
                pl = planes[0]
                idx = (d[pl] != 0).nonzero().flatten()
                yh_ = (y.reshape(n * c, -1)[pl][idx] * sc.reshape(-1)[pl] + sh.reshape(-1)[pl]).double()
                u_ = (gbuf.reshape(n * c, -1)[pl][idx].double() * torch.where(yh_ >= 0, 1.0, 0.2))
                A_ = torch.stack([u_, yh_, torch.ones_like(u_)], 1)
                dd_ = d[pl][idx].double()
                sol_ = torch.linalg.lstsq(A_, dd_.unsqueeze(1)).solution.flatten()
                res_ = (A_ @ sol_ - dd_).abs().max().item()
                print(f"   fit d = a u + b yh + c over the {idx.numel()} elements: a {sol_[0]:.3e} b {sol_[1]:.3e} c {sol_[2]:.3e}, residual {res_:.2e} (diffs up to {dd_.abs().max():.2e}); "
                      f"want {want.reshape(n * c, -1)[pl][idx[:4]].tolist()} got {out.reshape(n * c, -1)[pl][idx[:4]].tolist()}", flush=True)
                print(f"   launch {it}: {len(planes)} of {n * c} planes differ {planes[:8]}; plane {pl}: {idx.numel()} of {d.shape[1]} elements, first {int(idx[0])} last {int(idx[-1])}, "
                      f"512-thread slots hit {sorted(set(((idx // 4) % 512).tolist()))[:6]}.., k-slices {sorted(set(((idx // 4) // 512).tolist()))}, diffs {d[pl][idx[:4]].tolist()}", flush=True)
    print(f"{mode}{' wd' if 'wd' in sys.argv else ''}: convolutions beside = {beside}: {bad} of 200 act_bwd launches differ (worst {worst:.1e} of the largest value)", flush=True)
