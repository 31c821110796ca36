#!/bin/bash
# Developer aid: gpurun, retried while the pod has no free GPU slot (exit code 3: nothing ran, nothing was charged).
# usage: scripts/gpurun_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
