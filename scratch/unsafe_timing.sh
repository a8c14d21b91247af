one() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 20 2>/dev/null | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:round(d[k],2) for k in ('value','ms_per_step')}, d['config']['hip_graph_mode'])"; }
one SAN_UNSAFE_TIMING=0
one SAN_UNSAFE_TIMING=1
one SAN_UNSAFE_TIMING=2
one SAN_UNSAFE_TIMING=3
one SAN_NO_WGRAD_OVERLAP=1
