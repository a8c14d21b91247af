import csv, glob, sys
f = glob.glob("/tmp/pt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[1] if len(sys.argv) > 1 else "fft"
last = rows[-700:]
prev_end = None
out = []
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:30]
    if pat in name:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        out.append("%-30s dur=%7.1fus gap_before=%6.1fus grid=%s" % (name, (e - s) / 1e3, gap, r.get("Grid_Size_X", r.get("Workgroup_Size_X", "?"))))
    prev_end = e
print("\n".join(out[:44]))
