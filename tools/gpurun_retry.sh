#!/bin/bash
# gpurun, retried only while gpurun ITSELF says "no box / slot free right now" (its verdict
# `status=transient`: nothing ran, nothing charged).  The exit code alone cannot tell that from a command
# that really ran on the GPU box and exited 3 (pytest's internal error, bench.py's "needs an MI355X"), so the
# verdict gpurun writes to gpurun_out/.last_call.json is read instead; at most 12 tries.
# Usage: tools/gpurun_retry.sh [gpurun options] -- 'command'
here=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  status=$(python3 -c "import json,sys; print(json.load(open('$here/gpurun_out/.last_call.json')).get('status',''))" 2>/dev/null)
  [ "$status" = "transient" ] || exit $rc
  sleep 90
done
exit 3
