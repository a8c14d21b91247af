"""Timeline analysis of a kernel trace of replayed training steps (rocprofv3 --kernel-trace csv): per step the span, the time with
0 / 1 / >= 2 kernels running, per-queue busy time, and how long the weight-gradient stream runs beyond the main chain."""
import csv, glob, collections, sys
fs = glob.glob("/tmp/ptrain/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
# steps delimited by the second adamw launch of each step; take the last 4 steps
ends = idx[1::2]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
print("columns:", list(rows[0].keys()))
for si in range(len(ends) - 4, len(ends)):
    lo, hi = ends[si - 1] + 1, ends[si]
    seg = rows[lo:hi + 1]
    t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
    ev = []
    for r in seg:
        ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    cur = 0; last = t0; occ = collections.Counter()
    for t, d in ev:
        occ[min(cur, 2)] += t - last; last = t; cur += d
    perq = collections.defaultdict(lambda: [0, 0, 0])
    for r in seg:
        q = r[qkey] if qkey else "?"
        e = perq[q]; e[0] += 1; e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); e[2] = max(e[2], int(r["End_Timestamp"]))
    print(f"step {si}: span {1e-6 * (t1 - t0):.2f} ms, launches {len(seg)}; idle {1e-6 * occ[0]:.2f} ms, one kernel {1e-6 * occ[1]:.2f}, two or more {1e-6 * occ[2]:.2f}")
    for q, (c, d, e) in sorted(perq.items(), key=lambda kv: -kv[1][1]):
        print(f"    queue {q}: {c} launches, busy {1e-6 * d:.2f} ms, last end at +{1e-6 * (e - t0):.2f} ms")
    # gaps on the busiest queue
    main = max(perq.items(), key=lambda kv: kv[1][0])[0]
    ms = [r for r in seg if (r[qkey] if qkey else "?") == main]
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(ms, ms[1:])]
    gaps = [g for g in gaps if g > 0]
    print(f"    main-queue gaps: {len(gaps)} totalling {1e-6 * sum(gaps):.2f} ms, median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us")
