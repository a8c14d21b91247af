# host cores of a replayed step under runtime settings (bench.py prints host_cpu_ms = process CPU time per step)
one() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --main-only --no-kernel-timer --steps 20 2>/dev/null | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:round(d[k],2) for k in ('value','ms_per_step','host_enqueue_ms','host_cpu_ms')})"; }
one SAN_X=0
one SAN_NATIVE_REPLAY=0
one ROC_ACTIVE_WAIT_TIMEOUT=0
one ROC_ACTIVE_WAIT_TIMEOUT=100000
one HSA_ENABLE_INTERRUPT=0
one AMD_DIRECT_DISPATCH=0
one GPU_MAX_HW_QUEUES=2
one HIP_FORCE_DEV_KERNARG=1
