#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3; do
  SAN_B16_WD_COLD=0 SAN_B16_NBW=4 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "round-5 choices (direct weights, 256-pixel tiles):"
  SAN_B16_WD_COLD=1 SAN_B16_NBW=4 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "LDS-staged weights at <= 256 workgroups, 256-pixel tiles:"
  SAN_B16_WD_COLD=1 SAN_B16_NBW=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "LDS-staged weights at <= 256 workgroups, short tiles:"
done 2>&1 | tee gpurun_out/r6/wdcold_step.txt
timeout 1200 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
