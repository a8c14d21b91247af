"""Does RCCL come up with ONE rank on the 1-GPU box (communicator init + an in-place all-reduce on a side stream)?"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda:0")
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(1 << 20, device=dev, dtype=torch.float32)
dist.all_reduce(x); torch.cuda.synchronize()
print("init + first all-reduce %.2f s; backend %s; sum ok %s" % (time.time() - t0, dist.get_backend(), bool((x == torch.arange(1 << 20, device=dev)).all())))
s = torch.cuda.Stream()
big = torch.ones(30_000_000, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        e0.record(s); w = dist.all_reduce(big, async_op=True); w.wait(); e1.record(s)
    torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("120 MB single-rank all-reduce on a side stream: %.3f ms" % e0.elapsed_time(e1))
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        dist.all_reduce(big)
    g.replay(); torch.cuda.synchronize(); print("captured into a hipGraph and replayed")
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:200])
dist.destroy_process_group()
