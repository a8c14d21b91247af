"""Where do the +7 ms of the one-rank data-parallel step come from: RCCL being initialised, or the exchange path itself?"""
import os, sys, time, torch
os.environ.update(SAN_DIST_SINGLE="1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spatialalignmentnetwork_amd import dist as sdist, synth, ops
dev = torch.device("cuda:0")
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
net = bench.build_model(8, 320, 320, 12, dev); net.conv_dtype = "bf16x3"
run("no process group", net)
d = sdist.init("nccl", dev)
os.environ["SAN_DIST_SINGLE"] = "0"
net = bench.build_model(8, 320, 320, 12, dev); net.conv_dtype = "bf16x3"
run("RCCL group initialised, exchange path off", net)
os.environ["SAN_DIST_SINGLE"] = "1"
net = bench.build_model(8, 320, 320, 12, dev); net.conv_dtype = "bf16x3"
run("exchange path on (per cascade)", net)
net = bench.build_model(8, 320, 320, 12, dev); net.conv_dtype = "bf16x3"; net.time_exchange = True
run("exchange path on, time_exchange (event pairs around the collectives)", net)
print("allreduce_ms", net.exchange_ms() if hasattr(net, "exchange_ms") else None)
d.destroy_process_group()
