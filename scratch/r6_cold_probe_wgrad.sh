#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
for layer in 18-18-320-8 36-18-320-8 36-36-160-8 72-72-80-8 144-144-40-8 288-288-20-8; do
for cold in none xdy; do
  rm -rf /tmp/cp
  LAYER=$layer COLD=$cold timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o b --output-format csv -- python $R/scratch/r6_cold_probe_wgrad.py > /tmp/cp.txt 2>&1 < /dev/null
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/cp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wgrad' in r['Name'] and int(r['Calls']) >= 200:
            print('$layer COLD=$cold', r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:44], r['Calls'], 'avg %.1f us  min %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done; done 2>&1 | tee $R/gpurun_out/r6/cold_probe_wgrad.txt
