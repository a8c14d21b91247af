# FETCH_SIZE / WRITE_SIZE of the cascade-boundary launches at 15 x 640 x 368 (two --pmc passes) -> gpurun_out/pmc_dc_traffic.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for P in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$P
  timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pt_$P -o p -- python $R/scratch/bench_dc_rows.py 1 15 640 368 > /tmp/pt_$P.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in ('/tmp/pt_FETCH_SIZE', '/tmp/pt_WRITE_SIZE'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'dc_rows' not in k and 'coil_combine' not in k: continue
            e = acc[k[:70]][r['Counter_Name']]; e[0] += float(r['Counter_Value']); e[1] += 1
with open('$R/gpurun_out/pmc_dc_traffic.txt', 'w') as out:
    print('# 15 x 640 x 368, one sample: per-launch averages; FETCH_SIZE / WRITE_SIZE in KiB as reported (wide coalesced reads are tallied at half their bytes on gfx950: MI355X_MICROARCH.md, HBM section)', file=out)
    for k, d in acc.items():
        print(k, file=out)
        for c, (s, n) in sorted(d.items()): print(f'   {c:12s} {s / n:12.1f} KiB  ({n} launches)', file=out)
print(open('$R/gpurun_out/pmc_dc_traffic.txt').read())
PY
