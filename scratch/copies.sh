one() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 20 2>/dev/null | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:round(d[k],2) for k in ('value','ms_per_step')})"; }
for c in 4 6 8 12 16; do one SAN_DY_COPIES=$c; done
one SAN_DY_COPIES=8 SAN_WGRAD_BATCH=8
