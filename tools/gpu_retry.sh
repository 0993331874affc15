#!/bin/bash
# gpurun with patience: retries while the pod's GPU slots are busy (exit 3 = nothing charged).   tools/gpu_retry.sh TIMEOUT 'command'
T=$1; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    if [ $rc -ne 3 ] && ! grep -q '"status": "transient"' gpurun_out/.last_call.json 2>/dev/null; then exit $rc; fi
    sleep 90
done
exit 3
