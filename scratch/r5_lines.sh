#!/bin/bash
# refresh the bench lines only (no profiler)
TAG=${1:-r05}; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
python bench.py --dtype bf16 --no-cpu-baseline --main-only > gpurun_out/${TAG}_bench_line_bf16.json 2>/dev/null
python bench.py --dtype fp8 --no-cpu-baseline --main-only > gpurun_out/${TAG}_bench_line_fp8.json 2>/dev/null
python bench.py --mode infer --no-cpu-baseline --main-only > gpurun_out/${TAG}_bench_line_infer.json 2>/dev/null
SAN_DIST_SINGLE=1 python bench.py --no-cpu-baseline --main-only > gpurun_out/${TAG}_bench_line_rccl_one_rank.json 2>/dev/null
for nb in 1 2; do
python bench.py --no-cpu-baseline --coils 15 --height 640 --width 368 --sparsity 0.125 --batch $nb --steps 10 > gpurun_out/${TAG}_bench_line_config4_multicoil_n$nb.json 2>/dev/null
done
for f in gpurun_out/${TAG}_bench_line*.json; do echo $f; python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],3), (d.get('inference') or {}).get('value'))"; done
