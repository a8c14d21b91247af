#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
for layer in 144-288-20-8 288-288-20-8 288-144-20-8 36-72-80-8 72-72-80-8 144-72-80-8 72-36-80-8 144-72-40-8 144-288-40-8; do
for wd in -1 0; do
for cold in wx; do
  rm -rf /tmp/cp
  SAN_B16_WD=$wd LAYER=$layer COLD=$cold timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o b --output-format csv -- python $R/scratch/r6_cold_probe.py > /tmp/cp.txt 2>&1 < /dev/null
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/cp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if ('conv_bf16x3_kernel' in r['Name'] or 'conv3x3_stream' in r['Name'] or 'splitk' in r['Name']) and int(r['Calls']) >= 200:
            print('$layer WD=$wd COLD=$cold', r['Name'].split('::')[-1][:60].replace('(anonymous namespace)',''), r['Calls'], 'avg %.1f us  min %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done; done; done 2>&1 | tee $R/gpurun_out/r6/cold_probe3.txt
