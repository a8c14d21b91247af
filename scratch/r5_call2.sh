#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_hip_parity_r5.py -x -q 2>&1 | tail -25 > gpurun_out/r5/t5.txt
cat gpurun_out/r5/t5.txt
