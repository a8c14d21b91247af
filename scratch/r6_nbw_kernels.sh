#!/bin/bash
# per-kernel averages IN THE STEP with the default tiles only (SAN_B16_NBW=4) vs the short-tile form (automatic)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
for t in 4 0; do
  rm -rf /tmp/pb_$t
  (cd $R && SAN_B16_NBW=$t timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb_$t -o b --output-format csv -- python bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 10 --warmup 3 > /tmp/pb_$t.txt 2>&1 < /dev/null)
  grep '"metric"' /tmp/pb_$t.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NBW=$t', d['ms_per_step'])"
done
python - <<'PY' > $R/gpurun_out/r6/nbw_kernels.txt
import csv, glob, re
def load(t):
    out = {}
    for f in glob.glob(f'/tmp/pb_{t}/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            n = re.sub(r'\(.*', '', n)
            out[n] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return out
a, b = load('4'), load('0')
rows = []
for k in sorted(set(a) | set(b)):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
tot_a, tot_b = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print(f"total kernel ms: default {tot_a:.1f} short-auto {tot_b:.1f}")
for d, k, ca, ta, cb, tb in sorted(rows, key=lambda r: -abs(r[0]))[:30]:
    print(f"{d:+9.2f} ms  {k[:70]:70s} default {ca:6d} x {1e3*ta/max(ca,1):7.1f} us   short {cb:6d} x {1e3*tb/max(cb,1):7.1f} us")
PY
head -36 $R/gpurun_out/r6/nbw_kernels.txt
