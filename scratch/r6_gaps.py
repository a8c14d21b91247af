"""Main-queue gaps of one traced training step: every idle interval > 2 us with the kernels on both sides and what the other queues
were doing (rocprofv3 --kernel-trace CSV in /tmp/ptrain)."""
import csv, glob, re, collections
fs = glob.glob("/tmp/ptrain/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
ends = idx[1::2]
lo, hi = ends[-3] + 1, ends[-2]
seg = rows[lo:hi + 1]
main = collections.Counter(r[qkey] for r in seg).most_common(1)[0][0]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
ms = [r for r in seg if r[qkey] == main]
others = [r for r in seg if r[qkey] != main]
t0 = int(ms[0]["Start_Timestamp"])
tot = 0.0
print("step span %.2f ms" % ((int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6))
for a, b in zip(ms, ms[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g > 2.0:
        tot += g
        ga, gb = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
        busy = [short(r["Kernel_Name"]) for r in others if int(r["End_Timestamp"]) > ga and int(r["Start_Timestamp"]) < gb]
        print(f"+{(ga - t0) / 1e3:9.1f} us  gap {g:6.1f}  after {short(a['Kernel_Name'])[:42]:42s} before {short(b['Kernel_Name'])[:42]:42s} others: {','.join(sorted(set(busy)))[:80]}")
print("gaps > 2 us: %.2f ms" % (tot / 1e3))
