#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3; do
  for b in 48 24 12 8; do
    SAN_WGRAD_DEFER_BATCH=$b timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "deferred reductions per launch $b:"
  done
done 2>&1 | tee gpurun_out/r6/defer_batch.txt
