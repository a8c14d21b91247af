#!/bin/bash
# variants of the overlapped 15-coil step, repeated passes in one process (tests/test_hip_parity_r2.py SAN_MC_DIAG: scratch/attempts/r4_sens_overlap_debug.patch)
run() { echo "== $*"; for i in 1 2 3 4; do env "$@" SAN_MC_DIAG=1 python -m pytest tests/test_hip_parity_r2.py -q -s -k "multicoil_two_cascade_train" 2>&1 | grep -E "MCDIAG|Error" | sed -e 's/gradient tensors differ from the first pass//' -e 's/img_rec equal //' -e 's/MCDIAG rep//' | cut -c1-28 | paste -sd'|'; done; }
run A=0
