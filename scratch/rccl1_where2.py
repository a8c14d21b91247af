import os, sys, time, torch
os.environ.update(SAN_DIST_SINGLE="1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import dist as sdist
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
d = sdist.init("nccl", dev)              # FIRST, as bench.py does
from spatialalignmentnetwork_amd import synth, ops
def run(label, net, steps=20):
    xf, xa = (t.to(dev) for t in synth.phantom_pair(8, 1, 320, 320, seed=1234))
    net.train()
    for _ in range(5):
        net.set_input(xf, xa); net.update()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        net.set_input(xf, xa); net.update()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label:60s} {dt:6.2f} ms/step  ({net.step_mode[:6]})", flush=True)
ops.set_conv_precision("fp32")
mode = sys.argv[1] if len(sys.argv) > 1 else "on"
os.environ["SAN_DIST_SINGLE"] = "1" if mode == "on" else "0"
net = bench.build_model(8, 320, 320, 12, dev); net.conv_dtype = "bf16x3"
net.time_exchange = mode == "on"
run(f"RCCL initialised first; exchange path {mode}", net)
from spatialalignmentnetwork_amd import ops as o
print("side", o._WG["pool"], "comm", sdist.GradExchange._streams, flush=True)
d.destroy_process_group()
