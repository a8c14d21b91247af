#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_fft_dc.py -x -q 2>&1 | tail -4
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3; do
  SAN_DC_STATS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "train, plane_stats launch per cascade:"
  SAN_DC_STATS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "train, statistics from the boundary launch:"
  SAN_DC_STATS=0 timeout 600 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline --main-only 2>/dev/null | line "inference, plane_stats launch per cascade:"
  SAN_DC_STATS=1 timeout 600 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline --main-only 2>/dev/null | line "inference, statistics from the boundary launch:"
done 2>&1 | tee gpurun_out/r6/dc_stats.txt
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_step_runtime.py -x -q 2>&1 | tail -3
