#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for v in 0 1; do SAN_STREAM_SMALL_COUT=$v timeout 300 python scratch/r6_small_cout.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6/small_cout.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s')"; }
for i in 1 2 3; do
  SAN_STREAM_SMALL_COUT=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "18->3 data gradient on the direct fp32 kernel:"
  SAN_STREAM_SMALL_COUT=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | line "... on the persistent matrix-core kernel:"
done 2>&1 | tee -a gpurun_out/r6/small_cout.txt
