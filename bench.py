#!/usr/bin/env python3
"""Benchmark of the reconstruction + alignment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input already
resident in HBM: CSModel.set_input (fft2 -> column mask -> ifft2 -> rss),
forwardT (alignment U-Net + bilinear warp + smoothness loss) and forwardR
(VarNet: sensitivity net + 12 cascades of [ifft2.conj(S).sum -> NormUnet -> fft2
+ soft DC] + rss, then the SSIM loss), BASELINE.json configs[1]: batch 8 of
320x320 single-coil slices per GPU, 12 cascades, chans 18, sens_chans 8.
Slices are independent, so N GPUs run N shards with no data-path collective
(weak scaling); only the timing is reduced across ranks.

Rank 0 prints ONE JSON line (metric slices/s = all slices of all ranks / max
rank time) that also carries
  roofline      : the dominant kernel (by GPU time inside the timed region),
                  algorithmic work / HIP-event time of its launches (every 29th
                  launch of a kernel family is bracketed, run alone on the GPU:
                  an event pair around each of ~1,700 launches per step costs
                  ~6 % of the step and would serialise the two streams);
  roofline_fft_dc: the fused FFT + data-consistency kernels against HBM;
  cpu_baseline  : the CPU oracle (PyTorch CPU restatement of the reference) timed
                  on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3     # fp32 vector == fp32-input MFMA peak
BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak; the bf16x3 convolution executes 6 bf16 products per fp32 MAC


def build_model(n_per_gpu, h, w, num_cascades, dev, seed=0):
    from spatialalignmentnetwork_amd import synth
    from spatialalignmentnetwork_amd.basemodel import Config
    from spatialalignmentnetwork_amd.model import CSModel
    cfg = Config(sparsity=0.25, lr=1e-4, shape=w, coils=1, reg="Rec", mask="equispaced", weight_smooth=1000.0,
                 weight_gan=0.0, weight_gan_sim=0.0, weight_sim=1.0, use_amp=False, num_cascades=num_cascades)
    net = CSModel(cfg)
    net.net_mask.pruned = synth.equispaced_pruned(w, 0.25, 0)
    # random-init weights of the reference's architecture (no checkpoints offline), deterministic per name
    for sub, sd in (("net_T", 1), ("net_R", 2)):
        m = getattr(net, sub)
        m.load_state_dict(synth.fill_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=seed + sd))
    net.to(dev).eval()
    return net


def one_step(net, img_full, img_aux):
    with torch.no_grad():
        net.set_input(img_full, img_aux)
        net.loss_all = 0
        net.forwardT()
        net.forwardR()
    return net.img_rec


def train_step(net, img_full, img_aux):
    """set_input + update(): forward of T and R, hand-written backward, (RCCL grad all-reduce when
    world > 1,) AdamW for both networks -- the reference's 'Rec' regime (model.py:206-216)."""
    net.set_input(img_full, img_aux)
    net.update()


def usable_cores(cap=32):
    """Host threads the baseline may really use: the affinity mask and the cgroup CPU quota,
    not os.cpu_count() (a 256-thread oneDNN team on a quota-limited box crawls), capped."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(num_cascades, h, w, mode="infer", budget_s=15.0):
    """The oracle on host cores: same arithmetic as the reference's CPU path
    (same ATen kernels), N=1 slices one at a time until ~budget_s is spent.
    mode 'train' times forward + autograd backward of the 'Rec' objective (no optimiser step)."""
    from oracle import cpu_ref as O
    from spatialalignmentnetwork_amd import synth
    from spatialalignmentnetwork_amd.cross import SpatialTransformer
    from spatialalignmentnetwork_amd.varnet import VarNet
    cores = usable_cores()
    torch.set_num_threads(cores)
    t_shapes = [(k, tuple(v.shape)) for k, v in SpatialTransformer(1).state_dict().items()]
    r_shapes = [(k, tuple(v.shape)) for k, v in VarNet(num_cascades=num_cascades, use_ref=True).state_dict().items()]
    pT, pR = synth.fill_params(t_shapes, seed=1), synth.fill_params(r_shapes, seed=2)
    img_full, img_aux = synth.phantom_pair(1, 1, h, w, seed=1234)
    pruned = synth.equispaced_pruned(w, 0.25, 0)
    train = mode == "train"
    if train:
        for d in (pT, pR):
            for k, v in d.items():
                if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                    v.requires_grad_(True)

    def run():
        o = O.recon_align_forward(pT, pR, img_full, img_aux, pruned, shape=w, sparsity=0.25,
                                  num_cascades=num_cascades, training=train)
        if train:
            o["loss_all"].backward()

    warm = lambda: O.recon_align_forward(pT, pR, img_full, img_aux, pruned, shape=w, sparsity=0.25, num_cascades=1)
    with torch.set_grad_enabled(train):
        warm()                      # warm-up on a 1-cascade pass (thread pools, oneDNN primitives)
        t0 = time.perf_counter()
        done = 0
        while True:
            run()
            done += 1
            if time.perf_counter() - t0 > budget_s or done >= 256:
                break
        dt = time.perf_counter() - t0
    what = "forward + autograd backward (no optimiser step)" if train else "forward"
    return {"value": done / dt, "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"{done} slice(s) of the same workload ({what}), one at a time (N=1), 1 warm-up, "
                      f"{dt:.1f} s of CPU work"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="slices per GPU")
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--cascades", type=int, default=12)
    ap.add_argument("--mode", choices=["infer", "train"], default="train",
                    help="train (default): the full optimisation step incl. the gradient all-reduce; infer: forward only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step into a hipGraph and time replays (no per-kernel timer in that mode)")
    args = ap.parse_args()

    from spatialalignmentnetwork_amd import dist as sdist
    rank, local_rank, world = sdist.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = sdist.init("nccl", dev)      # RCCL; None when world == 1

    from spatialalignmentnetwork_amd import ops, synth
    n, h, w = args.batch, args.size, args.size
    net = build_model(n, h, w, args.cascades, dev)
    img_full, img_aux = synth.phantom_pair(n, 1, h, w, seed=1234 + rank)
    img_full, img_aux = img_full.to(dev), img_aux.to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    if args.mode == "train":
        net.train()
        step = lambda: train_step(net, img_full, img_aux)
    else:
        step = lambda: one_step(net, img_full, img_aux)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.graph:
        # the arena, packed weights, twiddles and masks exist after warm-up, so the step neither
        # allocates through the library nor synchronises: it is capture-safe
        graph = torch.cuda.CUDAGraph()
        cap_stream = torch.cuda.Stream()
        cap_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap_stream):
            one_step(net, img_full, img_aux)
            torch.cuda.current_stream().synchronize()
            with torch.cuda.graph(graph, stream=cap_stream):
                one_step(net, img_full, img_aux)
        torch.cuda.current_stream().wait_stream(cap_stream)
        graph.replay()
        torch.cuda.synchronize()
        step = graph.replay
        args.no_kernel_timer = True
    barrier()
    timer = None
    if not args.no_kernel_timer:
        timer = ops.KernelTimer()
        ops.TIMER = timer
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ops.TIMER = None
    dt = sdist.max_over_ranks(dt, dist, dev)

    infer = None
    if args.mode == "train":
        # the forward-only (serving) rate of the same model, timed right after, reported alongside
        net.eval()
        for _ in range(2):
            one_step(net, img_full, img_aux)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step(net, img_full, img_aux)
        torch.cuda.synchronize()
        barrier()
        dti = sdist.max_over_ranks(time.perf_counter() - t1, dist, dev)
        infer = {"value": n * world * args.steps / dti, "unit": "slices/s", "ms_per_step": 1e3 * dti / args.steps}
    if rank == 0:
        total_slices = n * world * args.steps
        out = {
            "metric": "slices/sec (320x320, 12-cascade VarNet+align)", "value": total_slices / dt, "unit": "slices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (convolutions from 16-18 channels up and all weight gradients: bf16 matrix cores, operands split in three, six products per MAC, fp32-equivalent; FFT / DC / norms / losses fp32)", "data": "synthetic",
            "config": {"workload": ("train step (regime Rec: fwd + hand-written bwd + grad all-reduce + AdamW) " if args.mode == "train" else "inference pass ") + f"set_input+align+warp+VarNet{args.cascades}+SSIM, "
                                   f"{n} slices/GPU of {h}x{w} single-coil, 4x equispaced mask, random-init weights",
                       "slices_per_gpu": n, "global_batch": n * world, "cascades": args.cascades,
                       "parallelism": f"dp{world} (independent slice shards, no data-path collective)",
                       "scalar_backend": sdist.BACKEND},
        }
        if infer is not None:
            out["inference"] = infer
        if timer is not None:
            tot = timer.totals()
            dom = max(tot, key=lambda k: tot[k]["ms"])
            # HBM bytes per launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE collected in two
            # separate rocprofv3 runs of this workload and corrected as profiles/r01_pmc_traffic.json states);
            # bench.py itself cannot run the profiler, so this is the profile's figure, not a live one
            pmc = {}
            try:
                pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                                  "r01_pmc_traffic.json")))["kernels"]
            except (OSError, KeyError, ValueError):
                pass
            for key, field in ((dom, "roofline"), ("fft_dc", "roofline_fft_dc"), ("conv3x3", "roofline_conv_fp32"),
                               ("conv3x3_bf16x3", "roofline_conv_bf16x3"), ("wgrad3x3_bf16x3", "roofline_wgrad_bf16x3"),
                               ("wgrad3x3", "roofline_wgrad_fp32")):
                if key not in tot or (field != "roofline" and key == dom):
                    continue
                d = tot[key]
                if d["sampled_launches"] == 0:
                    continue
                # rate over the bracketed launches (every 29th of the family); "ms" is that rate applied to all launches
                sec = d["ms"] * 1e-3
                extra = {}
                if key in ("conv3x3_bf16x3", "wgrad3x3_bf16x3"):
                    # algorithmic (fp32) FLOPs are what the layer needs; the kernel issues six bf16 products per MAC,
                    # so against the bf16 roof the matrix pipe sees 6x that
                    ach, peak, unit, bound = 6.0 * d["work"] / sec / 1e12, BF16_PEAK_TFLOPS, "TFLOP/s", "mfma"
                    extra = {"dtype": "bf16 (3-way split operands, 6 products per fp32 MAC)",
                             "algorithmic_tflops": d["work"] / sec / 1e12}
                elif d["unit"] == "FLOP":
                    ach, peak, unit, bound = d["work"] / sec / 1e12, FP32_PEAK_TFLOPS, "TFLOP/s", "mfma"
                else:
                    ach, peak, unit, bound = d["work"] / sec / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
                out[field] = {"kernel": key, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                              "frac": ach / peak,
                              "traffic": (pmc.get(key) or {}).get("hbm_bytes_per_launch") if args.mode == "train" and
                              (n, h, args.cascades) == (8, 320, 12) else None,
                              "traffic_source": "profiles/r01_pmc_traffic.json" if key in pmc else None,
                              "launches": d["launches"], "timed_launches": d["sampled_launches"],
                              "avg_launch_us": 1e3 * d["sampled_ms"] / d["sampled_launches"],
                              "share_of_step": d["ms"] / (1e3 * dt), **extra}
            out["kernel_ms_per_step"] = {k: v["ms"] / args.steps for k, v in tot.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cascades, h, w, args.mode)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
