#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_hip_parity_r5.py tests/test_hip_parity_r4.py -x -q -k "rccl or launcher" 2>&1 | tail -15
SAN_DIST_SINGLE=1 python bench.py --no-cpu-baseline --main-only --steps 10 2>gpurun_out/r5/rccl1.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['allreduce_ms'], d['config']['native_rccl'], d['config']['native_rccl_note'], d['config']['step_mode'])"
SAN_NATIVE_RCCL=0 SAN_DIST_SINGLE=1 python bench.py --no-cpu-baseline --main-only --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['allreduce_ms'], d['config']['native_rccl'], d['config']['native_rccl_note'])"
tail -5 gpurun_out/r5/rccl1.err
