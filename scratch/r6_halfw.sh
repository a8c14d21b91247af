#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -4
export BL_ONLY=36-72-80,72-72-80,144-72-80,72-36-80,72-144-80
for i in 1 2; do for v in 0 1; do echo "== SAN_B16_HALFW=$v"; SAN_B16_HALFW=$v timeout 300 python scratch/bench_layers.py conv 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r6/halfw_layers.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3; do
  SAN_B16_HALFW=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "32 x 8 tiles at 80^2:"
  SAN_B16_HALFW=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "40 x 6 tiles at 80^2:"
done 2>&1 | tee gpurun_out/r6/halfw_step.txt
