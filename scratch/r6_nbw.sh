#!/bin/bash
# short-tile form (three 16-pixel blocks per wave) of the tiled 3x3 kernel on the 40^2 / 20^2 levels: correctness with the form forced
# wherever it fits, per-layer A/B, whole-step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
SAN_B16_NBW=3 timeout 1200 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -4
export BL_ONLY=72-144-40,144-144-40,288-144-40,144-288-20,288-288-20,144-144-20,72-72-40
for i in 1 2; do
for v in 4 0 3; do
  echo "== SAN_B16_NBW=$v"
  SAN_B16_NBW=$v timeout 300 python scratch/bench_layers.py conv 2>&1 | grep -v amdgpu.ids
done; done 2>&1 | tee gpurun_out/r6/nbw_layers.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3; do
  SAN_B16_NBW=4 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "default tiles only:"
  SAN_B16_NBW=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "short tiles (auto):"
done 2>&1 | tee gpurun_out/r6/nbw_step.txt
