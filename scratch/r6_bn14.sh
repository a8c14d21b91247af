#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 900 python scratch/r6_act_bwd_cluster.py 2>&1 | grep -v amdgpu.ids | grep -i 'BN\|BatchNorm\|ok\|determin' | tee gpurun_out/r6/bn14_kernels.txt
timeout 1200 python -m pytest tests/test_norm_backward.py tests/test_gpu_norm.py -x -q 2>&1 | tail -3
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3; do
  SAN_BN_BWD_CLUSTER_MEMBERS=63 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "32 x 320^2 BatchNorm backward as three launches:"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "... as one cluster launch (14 float4 per thread):"
done 2>&1 | tee gpurun_out/r6/bn14_step.txt
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_step_runtime.py -x -q 2>&1 | tail -3
