import csv, glob
fs = glob.glob("/tmp/prof/**/*kernel_trace.csv", recursive=True)
if not fs:
    print("no kernel trace found"); raise SystemExit
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "")))
       for r in rows if "bf16x3" in r["Kernel_Name"] or "split_planes" in r["Kernel_Name"]]
g = [seq[i:i + 4] for i in range(0, len(seq), 4)]
for i in range(12, len(g), 13):
    print(" | ".join(f"{n.split('(')[0].split('::')[-1][:22]} {d / 1e3:.1f}us g={gs}" for n, d, gs in g[i]))
