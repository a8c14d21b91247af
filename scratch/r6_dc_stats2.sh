#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fft_dc.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for st in 0 1; do
rm -rf /tmp/pd
SAN_DC_STATS=$st timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pd -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 10 > /tmp/pd.txt 2>&1 < /dev/null
python - <<PY
import csv, glob
for f in glob.glob('/tmp/pd/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dc_rows320' in r['Name'] or 'plane_stats' in r['Name']:
            print('SAN_DC_STATS=$st', r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:40], r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
PY
done
