#!/bin/bash
# round 6, call 1: the cluster forms of the norm backward (correctness, timings), the backward tests, same-box step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 600 python scratch/r6_act_bwd_cluster.py > gpurun_out/r6/act_bwd_cluster.txt 2>&1; echo "cluster script rc=$?" >> gpurun_out/r6/act_bwd_cluster.txt
tail -40 gpurun_out/r6/act_bwd_cluster.txt
for i in 1 2; do
  for v in "0 0" "1 0" "1 1"; do
    set -- $v
    SAN_ACT_BWD_CLUSTER=$1 SAN_BN_BWD_CLUSTER=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --main-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('IN cluster $1 BN cluster $2:', d['ms_per_step'], 'ms', d['value'])"
  done
done 2>&1 | tee gpurun_out/r6/ab_cluster_step.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "bwd or backward or train_step or norm or alignment" 2>&1 | tail -5
