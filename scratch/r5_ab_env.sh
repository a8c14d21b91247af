#!/bin/bash
# same-box A/B of whole training steps: bash scratch/r5_ab_env.sh "<ENV=0 settings of arm B>" [reps]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
B="python bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 20 --warmup 3"
for i in $(seq 1 ${2:-2}); do
  $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A (default)   ', round(d['value'],2), round(d['ms_per_step'],3))"
  env $1 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B ($1)', round(d['value'],2), round(d['ms_per_step'],3))"
done
