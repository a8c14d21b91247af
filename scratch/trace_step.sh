# kernel trace of the bench's replayed steps; keeps the LAST full step as gpurun_out/r4b_step_trace.csv (start, end, queue, name)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptr
timeout 400 rocprofv3 --kernel-trace -d /tmp/ptr -o tr --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 6 > /tmp/ptr_stdout.txt 2>&1 < /dev/null
grep -v "rocprofv3\|amdgpu.ids" /tmp/ptr_stdout.txt | tail -12 | cut -c1-300
python - <<'PY'
import csv, glob, os
fs = glob.glob("/tmp/ptr/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seg = rows[-6000:]
t0 = int(seg[0]["Start_Timestamp"])
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4b_step_trace.csv")
with open(out, "w") as f:
    f.write("start_ns,end_ns,queue,name\n")
    for r in seg:
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:80].replace(",", ";")
        f.write(f"{int(r['Start_Timestamp']) - t0},{int(r['End_Timestamp']) - t0},{r['Queue_Id']},{nm}\n")
print(len(seg), "launches in the step")
PY
