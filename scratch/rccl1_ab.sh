one() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 20 2>/dev/null | grep '"metric"' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print({k:round(d[k],2) if d[k] is not None else None for k in ('value','ms_per_step','allreduce_ms','host_enqueue_ms')}, d['config']['collective_backend'])"; }
one SAN_X=0
one SAN_DIST_SINGLE=1
one SAN_DIST_SINGLE=1 SAN_GRAD_BUCKETS=single
one SAN_DIST_SINGLE=1 SAN_NATIVE_REPLAY=0
