#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "f16x2" 2>&1 | tail -8
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step:', d['ms_per_step'], 'ms', d['value'])"
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
