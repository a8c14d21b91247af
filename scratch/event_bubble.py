"""What does an event record / a cross-stream wait between two kernels of one stream cost the GPU?  (GPU-bound chain of ~25 us kernels)"""
import torch
dev = "cuda:0"
x = torch.zeros(96 << 20, device=dev); y = torch.zeros(96 << 20, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
N = 200
evs = [torch.cuda.Event() for _ in range(2 * N)]
def run(mode):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(N):
        x.add_(1.0)
        if mode >= 1:
            evs[i].record(main)
        if mode >= 2:
            side.wait_event(evs[i])
        if mode >= 3:
            with torch.cuda.stream(side):
                y.add_(1.0)
        if mode >= 4:
            evs[N + i].record(side)
            if i >= 4:
                main.wait_event(evs[N + i - 4])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3
for _ in range(2):
    for mode, name in enumerate(["kernels only", "+ event record on main after each", "+ side stream waits for it", "+ a kernel on the side stream", "+ side records, main waits for the one 4 back"]):
        print(f"{name:50s} {run(mode):7.2f} us per iteration", flush=True)
