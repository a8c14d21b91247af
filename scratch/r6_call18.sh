#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_norm_backward.py -x -q 2>&1 | tail -5
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2; do
  SAN_ACT_BWD_WAVE=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "plane kernel for small planes:"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "wave per plane:"
done 2>&1 | tee gpurun_out/r6/ab_wave_planes.txt
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
