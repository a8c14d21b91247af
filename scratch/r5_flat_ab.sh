#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
export BL_ONLY=72-144-40,144-144-40,288-144-40,144-288-20,288-288-20,144-72-80,72-72-80,36-72-80,64-64-160,128-64-160
for mode in conv; do
SAN_LIB_PATH=$PWD/scratch/ab/libsan_head.so python scratch/bench_layers.py $mode 2>/dev/null | grep -v ids > /tmp/o1.txt
python scratch/bench_layers.py $mode 2>/dev/null | grep -v ids > /tmp/n1.txt
SAN_LIB_PATH=$PWD/scratch/ab/libsan_head.so python scratch/bench_layers.py $mode 2>/dev/null | grep -v ids > /tmp/o2.txt
python scratch/bench_layers.py $mode 2>/dev/null | grep -v ids > /tmp/n2.txt
paste -d"|" <(cut -c1-46 /tmp/o1.txt) <(cut -c22-46 /tmp/n1.txt) <(cut -c22-46 /tmp/o2.txt) <(cut -c22-46 /tmp/n2.txt)
done
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_parity_r2.py -x -q 2>&1 | tail -5
