#!/bin/bash
# build + run variants of the v3 stream kernel on the GPU box (hipcc is there): phase breakdown per variant
for v in "" "-DSAN_V3_NODEAL_EPI" "-DSAN_V3_NODEAL_PF" "-DSAN_V3_NODEAL_EPI -DSAN_V3_NODEAL_PF"; do
  echo "=== variant: $v"
  SAN_EXTRA_HIPCC_FLAGS="-DSAN_STREAM_DBG $v" python -m spatialalignmentnetwork_amd.build --force > /dev/null 2>&1
  SAN_CONV_STREAM=3 BL_ONLY=${BL:-18-18-320} SC_CHECK=0 python scratch/stream_check.py 2>&1 | grep -v amdgpu.ids
  SAN_CONV_STREAM=3 python scratch/stream_dbg.py ${BL:-18-18-320} 2>&1 | grep -v amdgpu.ids
done
