#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r6/final_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a gpurun_out/r6/final_gpu_tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r6/final_bench_line.json
python -c "import json; d=json.loads(open('gpurun_out/r6/final_bench_line.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_fft_dc']['frac'], d['cpu_baseline']['value'])"
