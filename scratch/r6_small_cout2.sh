#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -5
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_step_runtime.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
