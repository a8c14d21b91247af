#!/bin/bash
# the FIRST overlapped backward of a fresh model, N processes: how often the alignment net's gradients differ from the serial pass
bad=0
for i in $(seq 1 ${1:-14}); do
  r=$(SAN_MC_DIAG=1 python -m pytest tests/test_hip_parity_r2.py -q -s -k "multicoil_two_cascade_train" 2>&1 | grep -E "MCDIAG rep 6" | sed -e 's/.*equal //' | cut -c1-12)
  echo "run $i: last (serial) pass vs first pass: $r"
done
