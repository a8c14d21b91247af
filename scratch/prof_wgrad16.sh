# per-kernel durations of the bf16x3 weight-gradient passes (rocprofv3 kernel trace), one line per layer
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o wb --output-format csv -- python /root/repo/scratch/bench_wgrad16.py > /tmp/prof_stdout.txt 2>&1 < /dev/null
find /tmp/prof -name "*.csv" < /dev/null
python - < /dev/null <<'PY'
PY
python /root/repo/scratch/prof_wgrad16.py < /dev/null
