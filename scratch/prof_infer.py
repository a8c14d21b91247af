import csv, glob, collections
fs = glob.glob("/tmp/pinf/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 5 of 8 passes: split by the rss kernel that ends a pass
idx = [i for i, r in enumerate(rows) if "ssim" in r["Kernel_Name"].lower() or "window_loss" in r["Kernel_Name"]]
per = max(1, len(idx) // 8)
lo = idx[-5 * per - 1] + 1
rows = rows[lo:idx[-1] + 1]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0]); busy = 0
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); busy += d
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    agg[k][0] += 1; agg[k][1] += d
print(f"kernels/pass {len(rows) / 5:.0f}  span/pass {span / 5e6:.2f} ms  sum(dur)/pass {busy / 5e6:.2f} ms")
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{d / 5e6:8.3f} ms  {c / 5:7.1f} calls  {d / c / 1e3:8.1f} us  {k}")
