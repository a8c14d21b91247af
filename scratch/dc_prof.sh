# kernel-level durations (rocprofv3 --kernel-trace --stats) of dc_rows at 15 x 640 x 368 for the libraries named in $1
# (scratch/libs/*.so; "name:ABL" runs the ablation variant with SAN_DC_ABL=ABL)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cp $R/spatialalignmentnetwork_amd/libsan_hip.so /tmp/keep.so
for spec in $1; do
  l=${spec%%:*}; abl=0; case $spec in *:*) abl=${spec##*:};; esac
  cp $R/scratch/libs/$l.so $R/spatialalignmentnetwork_amd/libsan_hip.so
  rm -rf /tmp/dcp
  SAN_DC_ABL=$abl timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dcp -o p --output-format csv -- python $R/scratch/bench_dc_rows.py ${2:-1 15 640 368} > /tmp/dcp.log 2>&1
  echo "== $spec: $(tail -1 /tmp/dcp.log | cut -c1-60)"
  python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/dcp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dc_rows' in r['Name'] or 'coil_combine' in r['Name']:
            print('   %-50s calls %s avg_us %.1f min_us %.1f' % (r['Name'].replace('(anonymous namespace)::', '')[:50], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
cp /tmp/keep.so $R/spatialalignmentnetwork_amd/libsan_hip.so
