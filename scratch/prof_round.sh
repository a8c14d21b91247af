# Round-2 profile set of the default bench workload (train step, N = 8, 320 x 320, 12 cascades), one call:
#   (1) rocprofv3 --kernel-trace --stats of `python bench.py`, as shipped, and of the 13 training steps alone with the
#       weight gradients in line (SAN_NO_WGRAD_OVERLAP=1 --main-only: the per-step kernel budget);
#   (2) PMC passes of scratch/pmc_traffic.py (calibration kernels + 3 train steps), each in its own run with
#       --kernel-trace only: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
# Results land in gpurun_out/r02_*; scratch/pmc_round_finalize.py turns the raw PMC json into profiles/r02_pmc.json.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
if [ "${2:-pmc}" != "pmc_only" ]; then
for mode in overlap serial; do
  rm -rf /tmp/pbench
  if [ $mode = serial ]; then export SAN_NO_WGRAD_OVERLAP=1; EXTRA=--main-only; else unset SAN_NO_WGRAD_OVERLAP; EXTRA=; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pbench -o b --output-format csv -- python $R/bench.py --no-cpu-baseline $EXTRA > /tmp/pbench_stdout.txt 2>&1 < /dev/null
  grep '"metric"' /tmp/pbench_stdout.txt | tail -1 > $R/gpurun_out/${TAG}_${mode}_bench_line.json
  for f in /tmp/pbench/*kernel_stats.csv /tmp/pbench/*/*kernel_stats.csv; do if [ -f "$f" ]; then cp "$f" $R/gpurun_out/${TAG}_${mode}_kernel_stats.csv; fi; done
done
fi
unset SAN_NO_WGRAD_OVERLAP
if [ "${2:-pmc}" = "pmc" ] || [ "${2:-pmc}" = "pmc_only" ]; then
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- python $R/scratch/pmc_traffic.py > /tmp/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o p -- python $R/scratch/pmc_traffic.py > /tmp/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_m -o p -- python $R/scratch/pmc_traffic.py > /tmp/pmc_m.log 2>&1
tail -2 /tmp/pmc_m.log
TAG=$TAG python - <<'PY'
import csv, glob, os, json
R = os.environ['GRAFT_REPO_ROOT']; TAG = os.environ['TAG']
def family(k):
    return ('cal_apply' if 'apply_kernel' in k else 'cal_rss' if 'rss_kernel' in k and 'bwd' not in k else
            'gemm1x1' if 'gemm1x1_f16_kernel' in k else 'conv_direct' if 'conv_direct_kernel' in k else
            'conv_bf16x3_1x1' if 'conv_bf16x3_kernel' in k and (', true, 1, ' in k or ', false, 1, ' in k) else
            'conv_bf16x3' if 'conv_bf16x3_kernel' in k else
            'conv_stream' if 'conv3x3_stream_kernel' in k else
            'act_bwd' if ('act_bwd' in k or 'bwd_stats_kernel' in k) else
            'wgrad_bf16x3' if 'wgrad_bf16x3_direct_kernel' in k else
            'wgrad1x1_bf16x3' if 'wgrad1x1_bf16x3_kernel' in k else
            'conv_mfma_3x3' if 'conv_mfma_kernel' in k and ', 3, ' in k else
            'conv_mfma_1x1' if 'conv_mfma_kernel' in k else
            'wgrad_vec_3x3' if 'conv_wgrad_vec_kernel<3' in k else
            'wgrad_vec_1x1' if 'conv_wgrad_vec_kernel<1' in k else
            'dc_rows' if 'dc_rows' in k else
            'fft320_rows' if 'fft320_rows' in k else 'fft320_cols' if 'fft320_cols' in k else None)
res = {}
for d in ('/tmp/pmc_f', '/tmp/pmc_w', '/tmp/pmc_m'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            key = family(r['Kernel_Name'])
            if key is None: continue
            tag = r['Counter_Name']
            if key.startswith('cal_'):
                res.setdefault(key, {}).setdefault(tag + '_list', []).append(float(r['Counter_Value']))
                continue
            e = res.setdefault(key, {}).setdefault(tag, [0.0, 0])
            e[0] += float(r['Counter_Value']); e[1] += 1
out = {k: {t: ({'sum': v[0], 'launches': v[1], 'avg': v[0] / v[1]} if not t.endswith('_list') else sorted(v, reverse=True)[:6]) for t, v in d.items()} for k, d in res.items()}
json.dump(out, open(f'{R}/gpurun_out/{TAG}_pmc_raw.json', 'w'), indent=1)
print(json.dumps({k: {t: (v if t.endswith('_list') else round(v['avg'], 1)) for t, v in d.items()} for k, d in out.items()})[:3000])
PY
fi
ls -la $R/gpurun_out/ | tail -8
