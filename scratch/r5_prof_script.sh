# usage: bash scratch/r5_prof_script.sh <out-name> <python script and args>   -- kernel stats of a script (avg us per kernel)
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rm -rf /tmp/ps
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ps -o p --output-format csv -- python $GRAFT_REPO_ROOT/"$@" > /tmp/ps.log 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r5/$name.txt
import csv, glob
for f in glob.glob('/tmp/ps/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'at::' not in r['Name']:
            print('%-90s calls %6s avg_us %8.1f min %8.1f' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ','')[:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs'])/1e3))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r5/$name.txt | head -40
