cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- python $R/scratch/pmc_traffic.py > /tmp/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o p -- python $R/scratch/pmc_traffic.py > /tmp/pmc_w.log 2>&1
tail -2 /tmp/pmc_f.log
python - <<'PY'
import csv, glob, os, json
R = os.environ['GRAFT_REPO_ROOT']
res = {}
for tag, d in (('FETCH_SIZE', '/tmp/pmc_f'), ('WRITE_SIZE', '/tmp/pmc_w')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != tag: continue
            k = r['Kernel_Name']
            key = ('cal_apply' if 'apply_kernel' in k else 'cal_rss' if 'rss_kernel' in k and 'bwd' not in k else
                   'conv_bf16x3_1x1' if 'conv_bf16x3_kernel' in k and (', true, 1, ' in k or ', false, 1, ' in k) else
                   'conv_bf16x3' if 'conv_bf16x3_kernel' in k else
                   'wgrad_bf16x3' if 'wgrad_bf16x3_direct_kernel' in k else
                   'wgrad1x1_bf16x3' if 'wgrad1x1_bf16x3_kernel' in k else
                   'conv_mfma_3x3' if 'conv_mfma_kernel' in k and ', 3, ' in k else
                   'conv_mfma_1x1' if 'conv_mfma_kernel' in k else
                   'wgrad_vec_3x3' if 'conv_wgrad_vec_kernel<3' in k else
                   'wgrad_vec_1x1' if 'conv_wgrad_vec_kernel<1' in k else
                   'fft320_rows' if 'fft320_rows' in k else 'fft320_cols' if 'fft320_cols' in k else None)
            if key is None: continue
            if key.startswith('cal_'):
                res.setdefault(key, {}).setdefault(tag + '_list', []).append(float(r['Counter_Value']))
                continue
            e = res.setdefault(key, {}).setdefault(tag, [0.0, 0])
            e[0] += float(r['Counter_Value']); e[1] += 1
out = {k: {t: ({'sum': v[0], 'launches': v[1], 'avg': v[0] / v[1]} if not t.endswith('_list') else sorted(v, reverse=True)[:6]) for t, v in d.items()} for k, d in res.items()}
os.makedirs(R + '/gpurun_out', exist_ok=True)
json.dump(out, open(R + '/gpurun_out/r01_pmc_traffic_raw.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith('cal_')}))
PY
