"""gpurun_out/<tag>_pmc_raw.json (scratch/prof_round.sh) -> profiles/<tag>_pmc.json.

HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are reported in KiB of 64-byte fabric requests; on gfx950 a wide coalesced
read is tallied at half its bytes (MI355X_MICROARCH.md, HBM section), other widths are uncalibrated -> the factors are
calibrated in the same run on kernels of known traffic far beyond the 256 MiB Infinity Cache (scratch/pmc_traffic.py:
apply_kernel 512 MiB in / out at 16 B and 4 B per lane, rss_kernel 512 MiB in at 8 B per lane).
MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES (SIMD-cycles with the matrix pipe busy, summed over the chip: 16 per 16x16x32 bf16
MFMA) / (elapsed cycles x 1024 SIMDs) per kernel family, elapsed = GRBM_GUI_ACTIVE / 8 (that counter comes back summed
over the 8 XCDs); SQ_BUSY_CYCLES and SQ_WAVE_CYCLES are kept next to it."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
raw = json.load(open(os.path.join(R, "gpurun_out", f"{tag}_pmc_raw.json")))
MiB = 1024.0 * 1024.0
cal = {}
ap = raw.get("cal_apply", {})
rs = raw.get("cal_rss", {})
# the two largest apply launches are the 16 B/lane shape (512 MiB), the next two the 4 B/lane shape (511.998 MiB)
f_list, w_list = ap.get("FETCH_SIZE_list", []), ap.get("WRITE_SIZE_list", [])
cal["raw_KiB"] = {"apply_FETCH": f_list, "apply_WRITE": w_list, "rss_FETCH": rs.get("FETCH_SIZE_list", [])}
# apply_kernel launches in descending raw order: the two 4 B / lane ones (511 x 513 planes: 511.998 MiB read and written), then
# the two 16 B / lane ones (512 x 512: 512 MiB); rss_kernel reads 512 MiB at 8 B / lane
f4 = 511.998 * MiB / (f_list[0] * 1024) if len(f_list) >= 4 else 2.0
f16 = 512 * MiB / (f_list[2] * 1024) if len(f_list) >= 4 else 2.0
f8 = 512 * MiB / (max(rs["FETCH_SIZE_list"]) * 1024) if rs.get("FETCH_SIZE_list") else f16
wf4 = 511.998 * MiB / (w_list[0] * 1024) if len(w_list) >= 4 else 1.0
wf = 512 * MiB / (w_list[2] * 1024) if len(w_list) >= 4 else 1.0
width = {"conv_mfma_3x3": f4, "conv_mfma_1x1": f4, "conv_bf16x3": f4, "conv_bf16x3_1x1": f4, "conv_stream": f4}
wwidth = {"conv_mfma_3x3": wf4, "conv_mfma_1x1": wf4, "conv_bf16x3": wf4, "conv_bf16x3_1x1": wf4}      # (the stream form stores 16 B per lane)
kern = {}
for k, v in raw.items():
    if k.startswith("cal_"):
        continue
    e = {}
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f = v["FETCH_SIZE"]["avg"] * 1024 * width.get(k, f16)
        w = v["WRITE_SIZE"]["avg"] * 1024 * wwidth.get(k, wf)
        e.update(launches=v["FETCH_SIZE"]["launches"], fetch_bytes_per_launch=int(f), write_bytes_per_launch=int(w),
                 hbm_bytes_per_launch=int(f + w), read_factor=round(width.get(k, f16), 3))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        busy, gui = v["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"], v["GRBM_GUI_ACTIVE"]["sum"]
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (the 1 GiB calibration copy reads 5.26 M "cycles" for ~310 us
        # at ~2.1 GHz), so elapsed cycles = gui / 8 and the chip offers (gui / 8) * 1024 SIMD-cycles
        e["mfma_busy"] = busy / (gui / 8.0 * 1024.0) if gui else None
        e["mfma_counters_avg_per_launch"] = {c: v[c]["avg"] for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
                                                                      "GRBM_GUI_ACTIVE") if c in v}
    kern[k] = e
# the bench's "conv3x3_bf16x3" family = the one-tile kernel + (round 4) the persistent stream kernel: launch-weighted merge
if "conv_stream" in kern and "conv_bf16x3" in kern and "hbm_bytes_per_launch" in kern["conv_stream"]:
    a_, b_ = kern["conv_bf16x3"], kern["conv_stream"]
    la, lb = a_["launches"], b_["launches"]
    m = {"launches": la + lb, "note": "one-tile kernel (conv_bf16x3) + persistent stream kernel (conv_stream), launch-weighted"}
    for key in ("fetch_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch"):
        m[key] = int((a_[key] * la + b_[key] * lb) / (la + lb))
    if a_.get("mfma_busy") is not None and b_.get("mfma_busy") is not None:
        ga, gb = a_["mfma_counters_avg_per_launch"]["GRBM_GUI_ACTIVE"] * la, b_["mfma_counters_avg_per_launch"]["GRBM_GUI_ACTIVE"] * lb
        m["mfma_busy"] = (a_["mfma_busy"] * ga + b_["mfma_busy"] * gb) / (ga + gb)
    kern["conv3x3_bf16x3_all"] = m
alias = {"conv3x3": "conv_mfma_3x3", "conv3x3_bf16x3": "conv3x3_bf16x3_all" if "conv3x3_bf16x3_all" in kern else "conv_bf16x3",
         "wgrad3x3_bf16x3": "wgrad_bf16x3", "wgrad3x3": "wgrad_vec_3x3"}
for a, b in alias.items():
    if b in kern:
        kern[a] = dict(kern[b])
if "dc_rows" in kern and "hbm_bytes_per_launch" in kern["dc_rows"]:
    kern["fft_dc"] = dict(kern["dc_rows"], note="the image-domain cascade kernel (forward launches with and without dk_out, and the backward form)")
out = {"note": "rocprofv3 --pmc passes of scratch/pmc_traffic.py (calibration kernels + 3 train steps at N = 8, 320 x 320, 12 cascades), "
               "each counter set in its own run with --kernel-trace only (scratch/prof_round.sh); corrected by scratch/pmc_round_finalize.py",
       "calibration": cal, "fetch_factor": {"16B_per_lane": f16, "8B_per_lane": f8, "4B_per_lane": f4}, "write_factor": {"16B_per_lane": wf, "4B_per_lane": wf4},
       "mfma_busy_definition": "SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs), summed over the family's launches",
       "kernels": kern}
json.dump(out, open(os.path.join(R, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps({k: {"hbm": v.get("hbm_bytes_per_launch"), "mfma_busy": v.get("mfma_busy")} for k, v in kern.items()}, indent=1))
