#!/bin/bash
# gpurun_retry.sh <timeout_s> <command...>: gpurun, retried every 90 s while the pod's GPU slots are busy (exit code 3) — at most 40 tries
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
