#!/bin/bash
# fresh processes, first launches of every stream-kernel form + 60 launches each: hang detection (an early build stalled at the
# first launch in 2 of 14 processes)
export SAN_CONV_STREAM=1
ok=0; bad=0
for i in $(seq 1 ${1:-30}); do
  if BL_ONLY=18-18-320,36-18-320,36-36-160,72-36-160,64-32-160,48-48-160 SC_CHECK=0 timeout 60 python scratch/stream_check.py > /tmp/stress_$i.txt 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); echo "process $i FAILED/HUNG"; tail -2 /tmp/stress_$i.txt; fi
done
echo "stream stress: $ok ok, $bad failed"
