# kernel-level durations of one layer's launches: bash scratch/layer_prof.sh "<libs>" <cin-cout-size> [conv|wgrad]
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cp $R/spatialalignmentnetwork_amd/libsan_hip.so /tmp/keep.so
for l in $1; do
  cp $R/scratch/libs/$l.so $R/spatialalignmentnetwork_amd/libsan_hip.so
  rm -rf /tmp/lp
  BL_ONLY=$2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/lp -o p --output-format csv -- python $R/scratch/bench_layers.py ${3:-wgrad} > /tmp/lp.log 2>&1
  echo "== $l: $(grep -E '^(conv|wgrad) ' /tmp/lp.log | tail -1)"
  python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/lp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r['Calls']) >= 20 and 'at::' not in r['Name']:
            print('   %-64s calls %s avg_us %.1f' % (r['Name'].replace('(anonymous namespace)::', '')[:64], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cp /tmp/keep.so $R/spatialalignmentnetwork_amd/libsan_hip.so
