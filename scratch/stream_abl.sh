#!/bin/bash
# same-box runs of scratch/stream_check.py over the stream kernel's tuning hooks
export SAN_CONV_STREAM=1
L=${BL_ONLY:-18-18-320,36-18-320,36-36-160,72-36-160}
run() { echo "== $*"; env "$@" BL_ONLY=$L SC_CHECK=0 timeout 60 python scratch/stream_check.py 2>&1 | grep -v amdgpu.ids; }
for i in 1 2 3; do echo "== check $i"; BL_ONLY=$L timeout 60 python scratch/stream_check.py 2>&1 | grep -v amdgpu.ids; done
run SAN_CONV_STREAM=0
run SAN_CONV_STREAM_STAG=0
for m in 0 2; do for d in 8 16 32 64 128; do run SAN_CONV_STREAM_STAGMODE=$m SAN_CONV_STREAM_STAG=$d; done; done
run SAN_CONV_STREAM_STAG=0
