#!/bin/bash
# gpurun, retried only while it answers "no box / slot free right now" (exit 3: nothing ran, nothing charged).
# Usage: tools/gpurun_retry.sh [gpurun options] -- 'command'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
