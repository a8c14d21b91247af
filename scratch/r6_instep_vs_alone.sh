#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
bash scratch/r6_gaps.sh > /dev/null 2>&1
cp gpurun_out/r6/trace_one_cascade.txt gpurun_out/r6/trace_one_cascade_b.txt
timeout 300 python scratch/bench_layers.py conv 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/layers_hot.txt
