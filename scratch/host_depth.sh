#!/bin/bash
# host CPU time per step of the replayed training step against the replay run-ahead throttle
for cfg in "0 3" "128 2" "256 2" "256 3" "512 2" "1024 2"; do
  set -- $cfg
  echo "== SAN_REPLAY_CHUNK=$1 SAN_REPLAY_LAG=$2"
  SAN_REPLAY_CHUNK=$1 SAN_REPLAY_LAG=$2 timeout 200 python bench.py --main-only --no-cpu-baseline --no-kernel-timer --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'slices/s', round(d['ms_per_step'],2), 'ms; host_enqueue_ms', round(d['host_enqueue_ms'],1), 'host_cpu_ms', round(d['host_cpu_ms'],1))"
done
