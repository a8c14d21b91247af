#!/bin/bash
# gr.sh <timeout> <command...>: gpurun with retries while the pod's GPU slots are busy
T=$1; shift
for i in $(seq 1 20); do
  out=$(gpurun --timeout $T -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit 0
done
echo "$out"; exit 3
