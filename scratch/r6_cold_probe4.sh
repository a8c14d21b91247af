#!/bin/bash
# cold-operand sweep of the launch-plan choices that were tuned with L2-hot operands: split-K parts, channel blocks per workgroup
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
run() {  # $1 = label, env already set
  rm -rf /tmp/cp
  COLD=wx timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o b --output-format csv -- python $R/scratch/r6_cold_probe.py > /tmp/cp.txt 2>&1 < /dev/null
  python - <<PY
import csv, glob
tot = 0.0; parts = []
for f in glob.glob('/tmp/cp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if ('conv_bf16x3_kernel' in r['Name'] or 'conv3x3_stream' in r['Name'] or 'splitk' in r['Name']) and int(r['Calls']) >= 200:
            tot += float(r['AverageNs']) / 1e3; parts.append('%.1f' % (float(r['AverageNs']) / 1e3))
print('$1  total %.1f us  (%s)' % (tot, ' + '.join(parts)))
PY
}
for layer in 144-288-20-8 288-288-20-8 288-144-20-8 288-144-40-8 144-72-40-8 144-288-40-8; do
  for sk in 1 2 4; do LAYER=$layer SAN_B16_SPLITK=$sk run "$layer split<=$sk"; done
done 2>&1 | tee $R/gpurun_out/r6/cold_probe4.txt
for layer in 36-72-80-8 72-72-80-8 144-72-80-8 72-36-80-8 72-144-80-8; do
  LAYER=$layer run "$layer auto"
  for mb in 2 3 4 5; do LAYER=$layer SAN_B16_MB=$mb run "$layer MB=$mb"; done
done 2>&1 | tee -a $R/gpurun_out/r6/cold_probe4.txt
