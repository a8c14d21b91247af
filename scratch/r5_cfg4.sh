#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
for nb in 1 2; do
python bench.py --no-cpu-baseline --main-only --coils 15 --height 640 --width 368 --sparsity 0.125 --batch $nb --steps 10 > gpurun_out/r5/cfg4_n$nb.json 2> gpurun_out/r5/cfg4_n$nb.err
python bench.py --no-cpu-baseline --main-only --mode infer --coils 15 --height 640 --width 368 --sparsity 0.125 --batch $nb --steps 10 > gpurun_out/r5/cfg4_infer_n$nb.json 2> gpurun_out/r5/cfg4_infer_n$nb.err
done
for f in gpurun_out/r5/cfg4_*.json; do echo $f; python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],2))"; done
