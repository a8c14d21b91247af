one() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 20 2>/dev/null | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:round(d[k],2) for k in ('value','ms_per_step')}, d['config']['hip_graph_mode'])"; }
for k in 1 2 4 8 16 32; do one SAN_WGRAD_BATCH=$k; done
one SAN_WGRAD_BATCH=1
