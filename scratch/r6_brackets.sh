#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
show() { python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
r=d.get('roofline') or {}
print('$1', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'slices/s | conv frac', round(r.get('frac',0),4), 'rocprof', round((r.get('rocprof') or {}).get('frac',0),4), '| fft_dc', round(d.get('roofline_fft_dc',{}).get('frac',0),4), round(d.get('roofline_fft_dc_bwd',{}).get('frac',0),4), '| wgrad', round(d.get('roofline_wgrad_bf16x3',{}).get('frac',0),4), '| act_bwd', round(d.get('roofline_act_bwd',{}).get('frac',0),4))"; }
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-only --no-kernel-timer 2>/dev/null | show "no brackets:"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-only --brackets-alone 2>/dev/null | show "brackets, launch alone (r2-r5):"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-only 2>/dev/null | show "brackets as shipped:"
done 2>&1 | tee gpurun_out/r6/brackets.txt
