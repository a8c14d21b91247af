#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 600 python scratch/r6_act_bwd_cluster.py 2>&1 | tail -8
for v in "0 0" "1 0" "0 1" "1 1"; do
  set -- $v
  echo "=== IN cluster $1, BN cluster $2"
  SAN_ACT_BWD_CLUSTER=$1 SAN_BN_BWD_CLUSTER=$2 timeout 600 python -m pytest tests/test_hip_parity_r2.py -x -q -s -k "test_train_step_full_320_golden" 2>&1 | grep "net_T\|passed\|failed"
done
