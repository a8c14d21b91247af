#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptrain
timeout 500 rocprofv3 --kernel-trace -d /tmp/ptrain -o tr --output-format csv -- python $GRAFT_REPO_ROOT/scratch/train_prof.py 6 > /tmp/ptrain_stdout.txt 2>&1 < /dev/null
tail -1 /tmp/ptrain_stdout.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6
python $GRAFT_REPO_ROOT/scratch/r6_gaps.py > $GRAFT_REPO_ROOT/gpurun_out/r6/gaps.txt 2>&1
(python $GRAFT_REPO_ROOT/scratch/trace_overlap.py < /dev/null | tail -8; python $GRAFT_REPO_ROOT/scratch/r5_trace_cascade.py < /dev/null) > $GRAFT_REPO_ROOT/gpurun_out/r6/trace_one_cascade.txt 2>&1
tail -5 $GRAFT_REPO_ROOT/gpurun_out/r6/gaps.txt
