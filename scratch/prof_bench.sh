# rocprofv3 --kernel-trace --stats of the default bench run; summary copied to gpurun_out/ (then to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pbench
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/pbench -o b --output-format csv -- python /root/repo/bench.py > /tmp/pbench_stdout.txt 2>&1 < /dev/null
mkdir -p /root/repo/gpurun_out
tail -1 /tmp/pbench_stdout.txt > /root/repo/gpurun_out/prof_bench_line.json
for f in /tmp/pbench/*kernel_stats.csv /tmp/pbench/*/*kernel_stats.csv; do if [ -f "$f" ]; then cp "$f" /root/repo/gpurun_out/prof_kernel_stats.csv; fi; done
ls -la /root/repo/gpurun_out/ | tail -5
head -5 /root/repo/gpurun_out/prof_kernel_stats.csv | cut -c1-150
