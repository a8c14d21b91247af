one() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 20 2>&1 | grep '"metric"' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print({k:round(d[k],2) for k in ('value','ms_per_step')})"; }
one SAN_SENS_OVERLAP=0
one SAN_SENS_DBG=1
one SAN_SENS_OVERLAP=0
one SAN_SENS_DBG=1
one SAN_SENS_DBG=0
