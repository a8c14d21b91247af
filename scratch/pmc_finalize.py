"""gpurun_out/r01_pmc_traffic_raw.json (scratch/pmc_traffic.sh) -> profiles/r01_pmc_traffic.json: apply the calibrated
FETCH_SIZE / WRITE_SIZE corrections and key the per-launch HBM bytes by bench.py's kernel names."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(os.path.join(R, "gpurun_out", "r01_pmc_traffic_raw.json")))
old = json.load(open(os.path.join(R, "profiles", "r01_pmc_traffic.json")))
F16, F4, WF = old["fetch_factor"]["16B_or_8B_per_lane"], old["fetch_factor"]["4B_per_lane"], old["write_factor"]
width = {"conv_mfma_3x3": "4B_per_lane", "conv_mfma_1x1": "4B_per_lane", "conv_bf16x3": "4B_per_lane", "conv_bf16x3_1x1": "4B_per_lane"}
kern = {}
for k, v in raw.items():
    if k.startswith("cal_") or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    w = width.get(k, "16B_or_8B_per_lane")
    f = v["FETCH_SIZE"]["avg"] * 1024 * (F4 if w == "4B_per_lane" else F16)
    wr = v["WRITE_SIZE"]["avg"] * 1024 * WF
    kern[k] = {"launches": v["FETCH_SIZE"]["launches"], "fetch_bytes_per_launch": int(f), "write_bytes_per_launch": int(wr),
               "hbm_bytes_per_launch": int(f + wr), "read_width": w,
               "raw_avg_KiB": {"FETCH_SIZE": v["FETCH_SIZE"]["avg"], "WRITE_SIZE": v["WRITE_SIZE"]["avg"]}}
alias = {"conv3x3": "conv_mfma_3x3", "conv3x3_bf16x3": "conv_bf16x3", "wgrad3x3_bf16x3": "wgrad_bf16x3", "wgrad3x3": "wgrad_vec_3x3"}
for a, b in alias.items():
    if b in kern:
        kern[a] = dict(kern[b])
if "fft320_rows" in kern and "fft320_cols" in kern:
    r, c = kern["fft320_rows"], kern["fft320_cols"]
    n = r["launches"] + c["launches"]
    kern["fft_dc"] = {"launches": n, "hbm_bytes_per_launch": int((r["hbm_bytes_per_launch"] * r["launches"] + c["hbm_bytes_per_launch"] * c["launches"]) / n),
                      "note": "launch-weighted average over the rows and (fused) columns kernels"}
if "conv_bf16x3" in kern:
    kern["conv_bf16x3"]["note"] = kern["conv3x3_bf16x3"]["note"] = ("activation loads are 4 B/lane, the packed-weight loads 16 B/lane: "
                                                                  "the 4 B factor is applied to all of FETCH_SIZE (upper bound)")
old["kernels"] = kern
old["calibration_raw_KiB"] = {k: v for k, v in raw.items() if k.startswith("cal_")}
old["note"] = old["note"].split(" (scratch/pmc_traffic.sh")[0] + (" (scratch/pmc_traffic.sh, workload scratch/pmc_traffic.py: calibration kernels + 3 train steps at "
              "N=8, 320x320, 12 cascades; taken after the bf16x3 weight-gradient kernels went in; finalised by scratch/pmc_finalize.py)." +
              old["note"].split("cascades; taken after the XCD-aware workgroup orders and the bf16x3 convolution went in).")[-1])
json.dump(old, open(os.path.join(R, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
print({k: v.get("hbm_bytes_per_launch") for k, v in kern.items()})
