# kernel-level times of the norm + LeakyReLU backward (stats + apply) at the U-Net's level sizes
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ab
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ab -o p --output-format csv -- python $R/scratch/bench_act_bwd.py > /tmp/ab.log 2>&1
cat /tmp/ab.log | grep act_bwd
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/ab/**/*kernel_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    import collections
    agg = collections.OrderedDict()
    for r in rows:
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
        if 'at::' in k: continue
        key = (k, r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''))
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += d
    for (k, g), (c, d) in agg.items():
        print('%-42s grid %-10s calls %4d avg %7.1f us' % (k, g, c, d / c / 1e3))
PY
