# rocprofv3 kernel stats of the narrow-precision train steps: bench.py --dtype bf16 / fp8, weight gradients in line
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export SAN_NO_WGRAD_OVERLAP=1
for dt in bf16 fp8; do
  rm -rf /tmp/pbench
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pbench -o b --output-format csv -- python $R/bench.py --dtype $dt --no-cpu-baseline --main-only > /tmp/pbench_stdout.txt 2>&1 < /dev/null
  grep '"metric"' /tmp/pbench_stdout.txt | tail -1 > $R/gpurun_out/${TAG}_${dt}_serial_bench_line.json
  for f in /tmp/pbench/*kernel_stats.csv /tmp/pbench/*/*kernel_stats.csv; do if [ -f "$f" ]; then cp "$f" $R/gpurun_out/${TAG}_${dt}_serial_kernel_stats.csv; fi; done
  tail -3 /tmp/pbench_stdout.txt | cut -c1-300
done
unset SAN_NO_WGRAD_OVERLAP
# the lines themselves, not under the profiler
for dt in bf16 fp8; do
  timeout 300 python $R/bench.py --dtype $dt --no-cpu-baseline --main-only 2>/dev/null | grep '"metric"' | tail -1 > $R/gpurun_out/${TAG}_${dt}_bench_line.json
  cut -c1-200 $R/gpurun_out/${TAG}_${dt}_bench_line.json
done
