#!/bin/bash
# full GPU suite + bench line; results under gpurun_out/r5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
(time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r5/suite_$1.txt 2>&1
python bench.py --no-cpu-baseline --main-only --steps 20 > gpurun_out/r5/bench_$1.json 2> gpurun_out/r5/bench_$1.err
tail -4 gpurun_out/r5/suite_$1.txt; python -c "import json; d=json.loads(open('gpurun_out/r5/bench_$1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
