#!/bin/bash
for i in $(seq 1 ${1:-6}); do
  SAN_MC_DIAG=1 python -m pytest tests/test_hip_parity.py tests/test_hip_parity_r2.py -m gpu -q -s 2>&1 | grep -E "MCDIAG|multi-coil net_T|failed|passed" | cut -c1-600
  echo "[run $i]"
done
