"""What the recorded step does between the last cascade's backward and the alignment network's backward: the tape entries (C-ABI
calls collapsed per stream, event records / waits spelled out) from the last san_normunet_bwd_tail on.  Looks for the cause of the
~1.1 ms main-queue gap in front of rss_bwd (profiles/r06_main_queue_gaps_after_fork_reorder.txt)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import synth, ops
dev = torch.device('cuda', 0)
n, h, w = 8, 320, 320
net = bench.build_model(n, h, w, 12, dev)
a, b = synth.phantom_pair(n, 1, h, w, seed=1234)
a, b = a.to(dev), b.to(dev)
net.train()
for _ in range(6): bench.train_step(net, a, b)
torch.cuda.synchronize()
print("mode", net.step_mode)
step = net._auto["step"]
calls = step.calls
names = []
streams = {torch.cuda.current_stream().cuda_stream: "main"}
def sname(h):
    if h not in streams: streams[h] = "s%d" % len(streams)
    return streams[h]
for fn, args, kind in calls:
    owner, name = getattr(fn, "__self__", None), getattr(fn, "__name__", str(fn))
    if isinstance(owner, torch.cuda.Event) and name == "record":
        names.append(("REC", "event %x on %s" % (id(owner) & 0xfffff, sname(args[0].cuda_stream))))
    elif isinstance(owner, torch.cuda.Stream) and name == "wait_event":
        names.append(("WAIT", "%s waits event %x" % (sname(owner.cuda_stream), id(args[0]) & 0xfffff)))
    elif isinstance(owner, torch.cuda.Stream) and name == "wait_stream":
        names.append(("WAIT", "%s waits stream %s" % (sname(owner.cuda_stream), sname(args[0].cuda_stream))))
    elif name.startswith("san_"):
        st = args[-1] if args and isinstance(args[-1], int) and args[-1] > 4096 else None
        names.append(("K", name, sname(st) if st is not None else "?"))
    else:
        names.append(("PY", name))
last = max(i for i, e in enumerate(names) if e[0] == "K" and e[1] == "san_normunet_bwd_tail")
first = next(i for i in range(last, len(names)) if names[i][0] == "K" and names[i][1] == "san_rss_bwd")
print("entries", len(names), "last tail at", last, "rss_bwd at", first)
run = None
for i in range(last - 30, min(len(names), first + 40)):
    e = names[i]
    if e[0] == "K":
        key = (e[2],)
        if run and run[0] == key: run[1].append(e[1]); continue
        if run: print("   %3d launches on %-5s %s ... %s" % (len(run[1]), run[0][0], run[1][0], run[1][-1]))
        run = [key, [e[1]]]
    else:
        if run: print("   %3d launches on %-5s %s ... %s" % (len(run[1]), run[0][0], run[1][0], run[1][-1])); run = None
        print("%6d %s %s" % (i, e[0], e[1]))
if run: print("   %3d launches on %-5s %s ... %s" % (len(run[1]), run[0][0], run[1][0], run[1][-1]))
