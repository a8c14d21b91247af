#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2; do
  (cd scratch/ab/tree_mid && timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "mid round 6 (fd62b12):")
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "current, weight scale on:"
  SAN_F16_WSCALE=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "current, weight scale off:"
done 2>&1 | tee gpurun_out/r6/ab_trees2.txt
for v in 1 0; do
echo "=== SAN_F16_WSCALE=$v"
SAN_F16_WSCALE=$v timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -s -k "e2e_full_320_golden or eval_bench_batch_n8 or train_step_full_320 or multicoil_640x368 or elementwise_on_shipped or conv_blocks_golden" 2>&1 | grep -i "rel\|err\|worst\|passed\|failed\|hip-vs" | cut -c1-260
done 2>&1 | tee gpurun_out/r6/wscale_e2e.txt
