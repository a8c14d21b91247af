#!/bin/bash
# scratch/grun.sh <timeout_s> <logfile> '<command>': gpurun with retries while every GPU slot of the pod is busy (exit 3: nothing charged)
T=$1; LOG=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@" > "$LOG" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "rc=$rc" >> "$LOG"; exit $rc; fi
  sleep 45
done
echo "rc=3 (gave up)" >> "$LOG"
