#!/bin/bash
# the 15-coil two-cascade golden step, repeated, per overlap variant: prints the T / R gradient errors of every run
for v in 0 1 2; do
  for i in $(seq 1 ${1:-8}); do
    SAN_SENS_DBG=$v python -m pytest tests/test_hip_parity_r2.py -q -s -k "multicoil_two_cascade_train" 2>&1 | grep -E "multi-coil net_T|failed|passed" | tr '\n' ' ' | cut -c1-230
    echo " [dbg=$v run $i]"
  done
done
