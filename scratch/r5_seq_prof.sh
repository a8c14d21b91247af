# usage: bash scratch/r5_seq_prof.sh <out-name> <python script and args>  -- kernels in launch order, consecutive launches of one kernel collapsed (count, median us)
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rm -rf /tmp/sq
timeout 600 rocprofv3 --kernel-trace -d /tmp/sq -o p --output-format csv -- python $GRAFT_REPO_ROOT/"$@" > /tmp/sq.log 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r5/$name.txt
import csv, glob, re
rows = list(csv.DictReader(open(glob.glob('/tmp/sq/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([\w:]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:40]
run, name = [], None
def flush():
    if run and 'rocclr' not in name and 'at::' not in name and 'Cijk' not in name:
        d = sorted(run)
        print('%-70s x%3d  median %7.1f us  min %7.1f' % (name, len(d), d[len(d) // 2], d[0]))
for r in rows:
    k = short(r['Kernel_Name'])
    if k != name:
        flush(); run, name = [], k
    run.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
flush()
PY
grep -v "pack_\|wscale\|fill" $GRAFT_REPO_ROOT/gpurun_out/r5/$name.txt | head -80
