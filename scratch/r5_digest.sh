#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --main-only --no-kernel-timer --digest"
for i in 1 2; do
$B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain ', d['state_digest'][:16], d['config']['step_mode'][:40])"
SAN_DIST_SINGLE=1 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single', d['state_digest'][:16], d['config']['step_mode'][:40], d['config']['native_rccl'])"
done
SAN_NATIVE_RCCL=0 SAN_DIST_SINGLE=1 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single torch', d['state_digest'][:16])"
SAN_GRAD_BUCKETS=single SAN_DIST_SINGLE=1 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single onebuffer', d['state_digest'][:16])"
SAN_AUTO_RECORD=0 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain eager', d['state_digest'][:16])"
SAN_AUTO_RECORD=0 SAN_DIST_SINGLE=1 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single eager', d['state_digest'][:16])"
