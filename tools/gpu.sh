#!/bin/bash
# developer: gpurun in the background + wait until the call has STARTED on its box (the tree is snapshotted when the box is acquired,
# not when gpurun is invoked: sources edited while the call queues travel half-edited with a stale library).
# usage: tools/gpu.sh <timeout_s> '<command>'   -> returns once the command runs; result later in gpurun_out/.last_call.json
T=$1; shift
rm -f gpurun_out/.last_call.json
nohup /usr/local/graft/bin/gpurun --timeout "$T" -- "$@" > /tmp/gpurun_last.log 2>&1 &
echo "gpurun pid $!"
for i in $(seq 1 120); do
  sleep 10
  if [ -f gpurun_out/.last_call.json ]; then echo "finished already"; tail -5 /tmp/gpurun_last.log; exit 0; fi
  if /usr/local/graft/bin/gpurun --status 2>/dev/null | grep -q '"elapsed_s"'; then echo "running on the box: the tree may be edited again"; exit 0; fi
done
echo "still queueing after 20 min"
