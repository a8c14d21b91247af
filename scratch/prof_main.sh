# rocprofv3 kernel stats of `bench.py --main-only` (the default training workload as shipped): kernels matching $1 (regex), and the line
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pm
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pm -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --main-only > /tmp/pm.log 2>&1
PAT="${1:-reduce}" python - <<'PY'
import csv, glob, os, re
pat = re.compile(os.environ['PAT'])
for f in glob.glob('/tmp/pm/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('total kernel ms', round(tot / 1e6, 1), 'calls', sum(int(r['Calls']) for r in rows))
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
        if pat.search(r['Name']):
            print('%-64s %6s %8.1f us %5.2f%%' % (r['Name'].replace('(anonymous namespace)::', '')[:64], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
grep '"metric"' /tmp/pm.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
