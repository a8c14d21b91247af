"""Per-family kernel time per step from profiles/r06_train_kernel_stats{,_serial}.csv (steps = adamw_dev_kernel calls / 2)."""
import csv, sys
def fam(k):
    k = k.replace("(anonymous namespace)::", "")
    if "wgrad_bf16x3_direct" in k: return "wgrad 3x3"
    if "wgrad1x1_bf16x3" in k: return "wgrad 1x1 / transposed"
    if "wgrad_bf16x3_reduce" in k or "wgrad_reduce_batch" in k: return "wgrad reductions"
    if "act_bwd" in k or "bwd_stats" in k or "bn_bwd" in k or "normunet_bwd" in k: return "norm + LeakyReLU backward (+ NormUnet head / tail)"
    if "conv3x3_stream" in k: return "conv 3x3 persistent (320^2 / 160^2)"
    if "conv_bf16x3_kernel" in k and ", 3, true," in k: return "conv 3x3 full-width tiles (20^2 / 40^2)"
    if "conv_bf16x3_kernel" in k: return "conv 3x3 32 x 8 tiles"
    if "gemm1x1" in k: return "1x1 / transposed GEMM"
    if "norm_finalize" in k or "splitk_reduce" in k: return "finalisers / split-K join"
    if "conv_direct" in k: return "direct fp32 conv"
    if any(t in k for t in ("avgpool", "add_kernel", "apply_kernel", "plane_stats", "upsample", "replicate", "partials", "bias_grad")): return "pool / add / apply / stats"
    return "FFT + DC, packing, AdamW, fp32 MFMA, losses, torch"
for tag, f in (("in line", "profiles/r06_train_kernel_stats_serial.csv"), ("as shipped", "profiles/r06_train_kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    steps = sum(int(r["Calls"]) for r in rows if "adamw_dev_kernel" in r["Name"]) / 2
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
    calls = sum(int(r["Calls"]) for r in rows) / steps
    agg = {}
    for r in rows:
        a = agg.setdefault(fam(r["Name"]), [0, 0.0])
        a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
    print(f"== {tag}: {steps:.0f} steps, {tot:.2f} ms of kernels per step, {calls:.0f} launches per step")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {c / steps:.0f} | {t / 1e6 / steps:.2f} | {100 * t / 1e6 / steps / tot:.1f} % | {t / c / 1e3:.1f} |")
