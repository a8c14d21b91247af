# rocprofv3 kernel stats of the config-4 bench (one 15-coil 640 x 368 slice): top kernels by total time
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p4
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o p --output-format csv -- python $R/bench.py --coils 15 --height 640 --width 368 --sparsity 0.125 --batch 1 --no-cpu-baseline --main-only > /tmp/p4.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/p4/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('total kernel ms', tot / 1e6, 'calls', sum(int(r['Calls']) for r in rows))
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:28]:
        print('%-64s %6s %8.1f us %5.1f%%' % (r['Name'].replace('(anonymous namespace)::', '')[:64], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
grep '"metric"' /tmp/p4.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['steps'], d['warmup'])"
