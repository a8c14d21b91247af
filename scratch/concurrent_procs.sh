cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --main-only --steps 40 --warmup 5 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
echo "single infer b8:"; run --mode infer --batch 8
echo "two concurrent infer b4:"; run --mode infer --batch 4 > /tmp/a.txt & run --mode infer --batch 4 > /tmp/b.txt; wait; cat /tmp/a.txt /tmp/b.txt
echo "two concurrent infer b8:"; run --mode infer --batch 8 > /tmp/a.txt & run --mode infer --batch 8 > /tmp/b.txt; wait; cat /tmp/a.txt /tmp/b.txt
echo "single train b8:"; run --batch 8
echo "two concurrent train b4:"; run --batch 4 > /tmp/a.txt & run --batch 4 > /tmp/b.txt; wait; cat /tmp/a.txt /tmp/b.txt
echo "two concurrent train b8:"; run --batch 8 > /tmp/a.txt & run --batch 8 > /tmp/b.txt; wait; cat /tmp/a.txt /tmp/b.txt
