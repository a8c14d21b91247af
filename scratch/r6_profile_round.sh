#!/bin/bash
# Round-6 evidence set on the CURRENT library (run last): kernel stats (overlapped = as shipped, and serial), PMC passes, bench lines.
#   bash scratch/r5_profile_round.sh [tag]     -> gpurun_out/<tag>_*
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in overlap serial; do
  rm -rf /tmp/pbench
  if [ $mode = serial ]; then export SAN_NO_WGRAD_OVERLAP=1 SAN_SENS_OVERLAP=0; else unset SAN_NO_WGRAD_OVERLAP SAN_SENS_OVERLAP; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pbench -o b --output-format csv -- python $R/bench.py --no-cpu-baseline --main-only --steps 20 > /tmp/pbench_stdout.txt 2>&1 < /dev/null
  grep '"metric"' /tmp/pbench_stdout.txt | tail -1 > $R/gpurun_out/${TAG}_${mode}_bench_line_under_rocprof.json
  for f in /tmp/pbench/*kernel_stats.csv /tmp/pbench/*/*kernel_stats.csv; do if [ -f "$f" ]; then cp "$f" $R/gpurun_out/${TAG}_${mode}_kernel_stats.csv; fi; done
done
unset SAN_NO_WGRAD_OVERLAP SAN_SENS_OVERLAP
bash $R/scratch/prof_round.sh $TAG pmc_only > /tmp/pr.log 2>&1 || true
cd $R
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
python bench.py --dtype bf16 --no-cpu-baseline --main-only > gpurun_out/${TAG}_bench_line_bf16.json 2>/dev/null
python bench.py --dtype fp8 --no-cpu-baseline --main-only > gpurun_out/${TAG}_bench_line_fp8.json 2>/dev/null
python bench.py --mode infer --no-cpu-baseline --main-only > gpurun_out/${TAG}_bench_line_infer.json 2>/dev/null
SAN_DIST_SINGLE=1 python bench.py --no-cpu-baseline --main-only > gpurun_out/${TAG}_bench_line_rccl_one_rank.json 2>/dev/null
SAN_NO_WGRAD_OVERLAP=1 python bench.py --no-cpu-baseline --main-only --no-kernel-timer > gpurun_out/${TAG}_bench_line_no_side_stream.json 2>/dev/null
python bench.py --no-cpu-baseline --main-only --no-kernel-timer > gpurun_out/${TAG}_bench_line_main_only.json 2>/dev/null
SAN_B16_WD_COLD=0 SAN_B16_NBW=4 python bench.py --no-cpu-baseline --main-only --no-kernel-timer > gpurun_out/${TAG}_bench_line_round5_conv_plans.json 2>/dev/null
for nb in 1 2; do
python bench.py --no-cpu-baseline --coils 15 --height 640 --width 368 --sparsity 0.125 --batch $nb --steps 10 > gpurun_out/${TAG}_bench_line_config4_multicoil_n$nb.json 2>/dev/null
done
for f in gpurun_out/${TAG}_*bench_line*.json; do echo $f; python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],3))"; done
bash scratch/r6_gaps.sh > /dev/null 2>&1
cp gpurun_out/r6/trace_one_cascade.txt gpurun_out/${TAG}_trace_one_cascade.txt
cp gpurun_out/r6/gaps.txt gpurun_out/${TAG}_main_queue_gaps.txt
ls -la gpurun_out | grep $TAG
