"""Would two half-batches on two streams beat one batch?  Two independent CSModel instances with N/2 slices each, each step
recorded (CSModel.record_update) under its own stream, replayed (a) one after the other, (b) call by call interleaved, against
one model with N slices.  Only a throughput probe: the two models do not share weights or gradients."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spatialalignmentnetwork_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "8"))
MODE = os.environ.get("MODE", "train")


def make(n, stream):
    net = bench.build_model(n, 320, 320, 12, dev)
    full, aux = (t.to(dev) for t in synth.phantom_pair(n, 1, 320, 320, seed=1234))
    with torch.cuda.stream(stream):
        if MODE == "train":
            net.train()
            rec = net.record_update(full, aux, warmup=1, restore=False)
        else:
            rec = net.record_forward(full, aux, warmup=1)
    torch.cuda.synchronize()
    return net, rec


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


s0 = torch.cuda.current_stream()
netF, recF = make(N, s0)
print(f"one model, N = {N}: {timeit(recF.replay):.2f} ms per step ({len(recF.calls)} calls)", flush=True)
del netF, recF
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
netA, recA = make(N // 2, sA)
netB, recB = make(N // 2, sB)


def serial():
    recA.replay()
    recB.replay()


def interleaved():
    a, b = recA.calls, recB.calls
    for i in range(max(len(a), len(b))):
        if i < len(a):
            a[i][0](*a[i][1])
        if i < len(b):
            b[i][0](*b[i][1])


import threading


def threaded():
    ta = threading.Thread(target=recA.replay)
    tb = threading.Thread(target=recB.replay)
    ta.start()
    tb.start()
    ta.join()
    tb.join()


print(f"two models, N = {N // 2} each, one after the other: {timeit(serial):.2f} ms per pair of steps", flush=True)
print(f"two models, N = {N // 2} each, calls interleaved on two streams: {timeit(interleaved):.2f} ms per pair of steps", flush=True)
print(f"two models, N = {N // 2} each, one replaying thread per model: {timeit(threaded):.2f} ms per pair of steps", flush=True)
