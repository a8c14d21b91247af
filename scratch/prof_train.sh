# kernel trace of a few train steps: busy fraction of the GPU and the top kernels (safe: no stdin reads)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptrain
timeout 400 rocprofv3 --kernel-trace -d /tmp/ptrain -o tr --output-format csv -- python /root/repo/scratch/train_prof.py 5 > /tmp/ptrain_stdout.txt 2>&1 < /dev/null
tail -2 /tmp/ptrain_stdout.txt
python /root/repo/scratch/prof_train.py < /dev/null
