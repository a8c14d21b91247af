#!/bin/bash
run() { echo "== $*"; for i in 1 2; do env "$@" SAN_MC_DIAG=1 python -m pytest tests/test_hip_parity_r2.py -q -s -k "multicoil_two_cascade_train" 2>&1 | grep -E "MCDIAG|Error" | sed -e 's/gradient tensors differ from the first pass//' -e 's/img_rec equal //' -e 's/MCDIAG rep//' | cut -c1-28 | paste -sd'|'; done; }
for k in 0 1 2 3 4; do run SAN_SENS_X=4 SAN_SENS_AGG_SET=$k; done
