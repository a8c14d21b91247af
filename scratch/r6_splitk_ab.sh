#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3; do
  SAN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/ab/lib_prev/libsan_hip.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "previous split-K rule:"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "cold-tuned split-K rule:"
done 2>&1 | tee gpurun_out/r6/splitk_step.txt
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
