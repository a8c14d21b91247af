"""One cascade of a traced training step, kernel by kernel: main-queue launches between two consecutive cascade-boundary launches
(forward: dc_rows320_kernel<0, backward: dc_rows320_kernel<1) with start offset, duration and the gap in front of each."""
import csv, glob, re, collections, sys
fs = glob.glob("/tmp/ptrain/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
ends = idx[1::2]
cands = [(ends[k - 1] + 1, ends[k]) for k in range(max(1, len(ends) - 3), len(ends))]
lo, hi = max(cands, key=lambda c: c[1] - c[0])
seg = rows[lo:hi + 1]
cnt = collections.Counter(r[qkey] for r in seg)
main = cnt.most_common(1)[0][0]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
ms = [r for r in seg if r[qkey] == main]
t0 = int(ms[0]["Start_Timestamp"])
for tag, which in (("forward", "dc_rows320_kernel<0"), ("backward", "dc_rows320_kernel<1")):
    pos = [i for i, r in enumerate(ms) if which in r["Kernel_Name"]]
    if len(pos) < 8:
        print(tag, "boundaries found:", len(pos)); continue
    a, b = pos[5], pos[6]
    span = (int(ms[b]["Start_Timestamp"]) - int(ms[a]["Start_Timestamp"])) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ms[a:b]) / 1e3
    print(f"==== {tag}: one cascade = {b - a} main-queue launches, span {span:.1f} us, kernel time {busy:.1f} us, gaps {span - busy:.1f} us")
    prev_end = int(ms[a - 1]["End_Timestamp"])
    for r in ms[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        wg = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
        print(f"  +{(s - int(ms[a]['Start_Timestamp'])) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:5.1f}  grid {g:>8} wg {wg:>4}  {short(r['Kernel_Name'])}")
        prev_end = e
    # side queues during the interval
    ta, tb = int(ms[a]["Start_Timestamp"]), int(ms[b]["Start_Timestamp"])
    side = [r for r in seg if r[qkey] != main and int(r["End_Timestamp"]) > ta and int(r["Start_Timestamp"]) < tb]
    print(f"  side queues in this interval: {len(side)} launches, busy {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in side) / 1e3:.1f} us")
# whole-step summary by phase of the main queue
print("step span %.2f ms; main queue busy %.2f ms in %d launches; other queues busy %.2f ms in %d launches" % (
    (max(int(r["End_Timestamp"]) for r in seg) - int(seg[0]["Start_Timestamp"])) / 1e6,
    sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ms) / 1e6, len(ms),
    sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if r[qkey] != main) / 1e6, len(seg) - len(ms)))
# the part of the step outside the cascades: before the first forward boundary and after the last backward boundary
pf = [i for i, r in enumerate(ms) if "dc_rows320_kernel<0" in r["Kernel_Name"]]
pb = [i for i, r in enumerate(ms) if "dc_rows320_kernel<1" in r["Kernel_Name"]]
for tag, lo_, hi_ in (("head (set_input, packing, alignment + sensitivity forward)", 0, pf[0]), ("tail (sensitivity + alignment backward, optimiser)", pb[-1], len(ms))):
    seg_ = ms[lo_:hi_]
    print(f"==== {tag}: {len(seg_)} launches, {(int(seg_[-1]['End_Timestamp']) - int(seg_[0]['Start_Timestamp'])) / 1e3:.1f} us")
    agg = collections.OrderedDict()
    for r in seg_:
        k = short(r["Kernel_Name"])
        c, t = agg.get(k, (0, 0))
        agg[k] = (c + 1, t + int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"   {t / 1e3:8.1f} us  x{c:3d}  avg {t / c / 1e3:6.1f}  {k}")
