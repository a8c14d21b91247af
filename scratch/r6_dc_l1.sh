#!/bin/bash
# A/B: dc_rows320 with 2 lines per wave (shipped) vs 1 line per wave (2,560 waves at N = 8: 2.5 per SIMD)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for i in 1 2; do
for L in 2 1; do
  echo "== SAN_DC_L320=$L"
  SAN_DC_L320=$L DC_TRAIN=1 timeout 300 python scratch/bench_dc_rows.py 8 1 320 320
  SAN_DC_L320=$L DC_TRAIN=1 timeout 300 python scratch/bench_dc_rows.py 1 15 320 320
done; done 2>&1 | tee gpurun_out/r6/dc_l1.txt
SAN_DC_L320=1 timeout 900 python -m pytest tests/test_gpu_fft_dc.py -x -q 2>&1 | tail -3 | tee -a gpurun_out/r6/dc_l1.txt
cd /tmp && export TMPDIR=/tmp
for L in 2 1; do
  SAN_DC_L320=$L rocprofv3 --kernel-trace --stats -d /tmp/prof_dc$L -o dc -- python $GRAFT_REPO_ROOT/scratch/bench_dc_rows.py 8 1 320 320 > /dev/null 2>&1
  echo "== rocprof L=$L"; python - <<PY
import csv,glob
f=glob.glob('/tmp/prof_dc$L/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'dc_rows' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3,'us')
PY
done 2>&1 | tee -a $GRAFT_REPO_ROOT/gpurun_out/r6/dc_l1.txt
