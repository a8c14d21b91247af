#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3; do
  SAN_UNET_SHARE_GRADS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "copy per reader:"
  SAN_UNET_SHARE_GRADS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "shared gradient:"
done 2>&1 | tee gpurun_out/r6/share_grads.txt
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_step_runtime.py tests/test_gpu_warp_loss.py -x -q 2>&1 | tail -3
python - <<'PY'
# bit-identity of the two forms: parameters after 3 steps
import os, sys, torch, subprocess, json
PY
