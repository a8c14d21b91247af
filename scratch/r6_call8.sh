#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_norm_forward.py -x -q 2>&1 | tail -5
for i in 1 2; do
  for v in 0 1; do
    SAN_FUSED_FIN_POOL=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('finalise + pool in one launch $v:', d['ms_per_step'], 'ms', d['value'])"
  done
done 2>&1 | tee gpurun_out/r6/ab_fin_pool.txt
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
