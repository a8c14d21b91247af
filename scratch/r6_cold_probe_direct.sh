#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
for kind in dgrad fwd out; do
for cold in none x; do
  rm -rf /tmp/cp
  KIND=$kind COLD=$cold timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o b --output-format csv -- python $R/scratch/r6_cold_probe_direct.py > /tmp/cp.txt 2>&1 < /dev/null
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/cp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r['Calls']) >= 200 and 'conv' in r['Name']:
            print('$kind COLD=$cold', r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:40], r['Calls'], 'avg %.1f us  min %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done; done 2>&1 | tee $R/gpurun_out/r6/cold_probe_direct.txt
