import csv, glob, collections
fs = glob.glob("/tmp/ptrain/**/*kernel_trace.csv", recursive=True)
if not fs:
    print("no kernel trace found"); raise SystemExit
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 5 steps: find AdamW launches as step delimiters
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
per = len(idx) // 8                     # 3 warm-up + 5 timed steps
lo = idx[-5 * per - 1] + 1
rows = rows[lo:idx[-1] + 1]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = 0; last_end = int(rows[0]["Start_Timestamp"]); gaps = 0
agg = collections.defaultdict(lambda: [0, 0])
gap_by = collections.defaultdict(int); prev_end = int(rows[0]['Start_Timestamp'])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if s > last_end: gaps += s - last_end
    last_end = max(last_end, e)
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    gap_by[k] += max(0, s - prev_end)
    prev_end = max(prev_end, e)
    agg[k][0] += 1; agg[k][1] += e - s
steps = 5
print(f"kernels/step {len(rows) / steps:.0f}  span/step {span / steps / 1e6:.2f} ms  sum(dur)/step {busy / steps / 1e6:.2f} ms  idle gaps/step {gaps / steps / 1e6:.2f} ms")
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{d / steps / 1e6:8.3f} ms  {c / steps:7.1f} calls  {d / c / 1e3:8.1f} us  gap-before {gap_by[k] / steps / 1e6:6.3f} ms  {k}")
