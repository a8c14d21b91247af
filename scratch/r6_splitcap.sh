#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3 4; do
  SAN_B16_SPLITCAP=1024 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "split cap 1024 (288->288 @20^2 in four parts):"
  SAN_B16_SPLITCAP=512 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "split cap 512 (two parts):"
done 2>&1 | tee gpurun_out/r6/splitcap.txt
