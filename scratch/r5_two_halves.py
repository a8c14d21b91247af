"""Experiment (round 5): does the chip run two independent half-batch steps (N = 4 each, own streams, replayed from two host
threads) faster than one N = 8 step?  If yes, the cascades (per-sample normalisation: samples are independent) could be issued
as two interleaved half-batch chains that fill each other's dependent-boundary bubbles and kernel tails."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spatialalignmentnetwork_amd import synth, ops

dev = torch.device("cuda:0")
H = W = 320


def make(n, stream, seed):
    with torch.cuda.stream(stream):
        net = bench.build_model(n, H, W, 12, dev, seed=seed).train()
        full, aux = (t.to(dev) for t in synth.phantom_pair(n, 1, H, W, seed=1234 + seed))
        net.auto_record = False
        step = net.record_update(full, aux, warmup=2, restore=True)
    torch.cuda.synchronize()
    return net, step


def timeit(fn, iters=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


s0 = torch.cuda.current_stream()
net8, step8 = make(8, s0, 0)
t8 = timeit(step8.replay)
print(f"one N = 8 step                      : {t8:7.2f} ms  ({8e3 / t8:6.1f} slices/s)", flush=True)
del net8, step8
torch.cuda.empty_cache()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
netA, stepA = make(4, sA, 10)
tA = timeit(stepA.replay)
print(f"one N = 4 step                      : {tA:7.2f} ms  ({4e3 / tA:6.1f} slices/s)", flush=True)
netB, stepB = make(4, sB, 20)


def both_seq():
    stepA.replay()
    stepB.replay()


def both_threads():
    ta = threading.Thread(target=stepA.replay)
    tb = threading.Thread(target=stepB.replay)
    ta.start(); tb.start(); ta.join(); tb.join()


t2s = timeit(both_seq)
print(f"two N = 4 steps, one host thread    : {t2s:7.2f} ms  ({8e3 / t2s:6.1f} slices/s)", flush=True)
t2t = timeit(both_threads)
print(f"two N = 4 steps, two host threads   : {t2t:7.2f} ms  ({8e3 / t2t:6.1f} slices/s)", flush=True)
