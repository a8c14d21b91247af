"""Does an event WITHOUT the system-scope fence (hipEventDisableSystemFence) cost the main stream less than torch's default event?
Chain of ~25 us kernels on the main stream with an event record after each (and a side stream waiting for it), raw HIP calls."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
dev = "cuda:0"
x = torch.zeros(96 << 20, device=dev); y = torch.zeros(96 << 20, device=dev)
side = torch.cuda.Stream(); main = torch.cuda.current_stream()
N = 200
DISABLE_TIMING, DISABLE_SYSTEM_FENCE, RELEASE_TO_DEVICE = 0x2, 0x20000000, 0x40000000


def make(flags):
    evs = []
    for _ in range(N):
        e = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
        evs.append(e)
    return evs


def run(evs, wait):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ms, ss = ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(side.cuda_stream)
    for i in range(N):
        x.add_(1.0)
        if evs is not None:
            assert hip.hipEventRecord(evs[i], ms) == 0
            if wait:
                assert hip.hipStreamWaitEvent(ss, evs[i], 0) == 0
                with torch.cuda.stream(side):
                    y.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


sets = {"no events": None, "default flags (timing off)": make(DISABLE_TIMING), "DisableSystemFence": make(DISABLE_TIMING | DISABLE_SYSTEM_FENCE),
        "ReleaseToDevice": make(DISABLE_TIMING | RELEASE_TO_DEVICE)}
for _ in range(2):
    for name, evs in sets.items():
        print(f"{name:32s} record only {run(evs, False):7.2f} us / iteration   record + side wait + side kernel {run(evs, True):7.2f}", flush=True)
