#!/bin/bash
# same-box A/B: one vs two (vs three) weight-gradient side streams, interleaved; bit-identity of the step's parameters after 3 steps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2; do
  for k in 1 2 3; do
    SAN_WGRAD_STREAMS=$k timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "streams=$k:"
  done
  SAN_WGRAD_STREAMS=2 GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "streams=2 hwq=8:"
  SAN_WGRAD_STREAMS=2 SAN_WGRAD_BATCH=2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "streams=2 batch=2:"
done 2>&1 | tee gpurun_out/r6/wgstreams.txt
SAN_WGRAD_STREAMS=2 timeout 1500 python -m pytest tests/test_gpu_step_runtime.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -5 | tee -a gpurun_out/r6/wgstreams.txt
