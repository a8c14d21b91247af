#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
(time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) 2>&1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --main-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('rocprof'), d['roofline_fft_dc']['frac'], d['roofline_fft_dc'].get('rocprof'))"
