cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pinf
timeout 400 rocprofv3 --kernel-trace -d /tmp/pinf -o tr --output-format csv -- python /root/repo/scratch/infer_prof.py > /tmp/pinf_stdout.txt 2>&1 < /dev/null
grep "wall ms" /tmp/pinf_stdout.txt
python /root/repo/scratch/prof_infer.py < /dev/null
