#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, sys
sys.path.insert(0, '.')
from spatialalignmentnetwork_amd import ops
import torch.nn.functional as F
dev='cuda:0'; torch.manual_seed(0)
n,cin,cout,h,w=8,32,2,320,320
x=torch.randn(n,cin,h,w,device=dev); wt=torch.randn(cout,cin,3,3,device=dev)*0.1; b=torch.randn(cout,device=dev)
for on in (False, True):
    ops.STREAM_SMALL_COUT[0]=on
    y=torch.empty(n,cout,h,w,device=dev)
    ops.conv2d(ops.full(x), wt, b, ops.full(y))
    torch.cuda.synchronize()
    ref=F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    for _ in range(5): ops.conv2d(ops.full(x), wt, b, ops.full(y))
    e0.record()
    for _ in range(50): ops.conv2d(ops.full(x), wt, b, ops.full(y))
    e1.record(); torch.cuda.synchronize()
    print('small', on, 'err %.2e' % ((y.double()-ref).norm()/ref.norm()).item(), 'us %.1f' % (e0.elapsed_time(e1)*1e3/50))
PY
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_conv.py -x -q 2>&1 | tail -3
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2; do
  timeout 600 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline --main-only 2>/dev/null | line "inference:"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "train:"
done
