#!/bin/bash
# fresh processes, repeated: the overlapped-vs-serial full-size tests, the launcher test, the new kernels' tests
cd $GRAFT_REPO_ROOT
ok=0; bad=0
for i in $(seq 1 ${1:-6}); do
  if timeout 900 python -m pytest tests/test_hip_parity_r5.py -x -q -k "overlapped or failed_recording or handoff or repeat" > /tmp/st_$i.txt 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); tail -5 /tmp/st_$i.txt; fi
done
echo "stress: $ok ok, $bad failed"
timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -3
