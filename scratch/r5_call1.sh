#!/bin/bash
# round 5, GPU call 1: packed-fp32 reproducer + A/B of the library built without packed fp32
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 300 ./scratch/probe/pk32_repro 200 > gpurun_out/r5/pk32_repro.txt 2>&1
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
for i in 1 2; do
$B --main-only > gpurun_out/r5/ab_base_$i.json 2> gpurun_out/r5/ab_base_$i.err
SAN_LIB_PATH=$PWD/scratch/ab/libsan_nopk.so $B --main-only > gpurun_out/r5/ab_nopk_$i.json 2> gpurun_out/r5/ab_nopk_$i.err
done
SAN_LIB_PATH=$PWD/scratch/ab/libsan_nopk.so python scratch/bench_dc_rows.py > gpurun_out/r5/dc_nopk.txt 2>&1
python scratch/bench_dc_rows.py > gpurun_out/r5/dc_base.txt 2>&1
SAN_LIB_PATH=$PWD/scratch/ab/libsan_nopk.so python scratch/bench_fft.py > gpurun_out/r5/fft_nopk.txt 2>&1
python scratch/bench_fft.py > gpurun_out/r5/fft_base.txt 2>&1
SAN_LIB_PATH=$PWD/scratch/ab/libsan_nopk.so python scratch/bench_layers.py > gpurun_out/r5/layers_nopk.txt 2>&1
python scratch/bench_layers.py > gpurun_out/r5/layers_base.txt 2>&1
tail -3 gpurun_out/r5/pk32_repro.txt
for f in gpurun_out/r5/ab_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
