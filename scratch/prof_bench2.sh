# rocprofv3 --kernel-trace --stats of the default bench run, (1) as shipped (weight gradients overlapped on a side stream)
# and (2) with SAN_NO_WGRAD_OVERLAP=1 (serial: per-kernel durations without contention)
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out
for mode in overlap serial; do
  rm -rf /tmp/pbench
  if [ $mode = serial ]; then export SAN_NO_WGRAD_OVERLAP=1; fi
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/pbench -o b --output-format csv -- python /root/repo/bench.py --no-cpu-baseline > /tmp/pbench_stdout.txt 2>&1 < /dev/null
  grep '"metric"' /tmp/pbench_stdout.txt | tail -1 > /root/repo/gpurun_out/prof_${mode}_bench_line.json
  for f in /tmp/pbench/*kernel_stats.csv /tmp/pbench/*/*kernel_stats.csv; do if [ -f "$f" ]; then cp "$f" /root/repo/gpurun_out/prof_${mode}_kernel_stats.csv; fi; done
done
ls -la /root/repo/gpurun_out/ | tail -6
