#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 20 --warmup 3"
for rep in 1 2; do
for cfg in "X=0" "SAN_WGRAD_BATCH=2" "SAN_WGRAD_BATCH=3" "SAN_WGRAD_BATCH=6" "SAN_DY_COPIES=12" "SAN_WGRAD_BATCH=2 SAN_DY_COPIES=12"; do
  env $cfg $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value'],2), round(d['ms_per_step'],3))"
done; done
