#!/bin/bash
# video frames/s with and without 8 busy host threads beside the bench process (and with / without the two-stage pipeline)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/hostload
mkdir -p $O
cd $R
run() { python3 bench.py --mode video --frames 200 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']['rank0_clip']
print('$1', round(d['value'],1), 'frames/s; frontend', round(200/c['frontend_seconds'],1))"; }
W=${1:-584}; H=${2:-328}
for rep in 1 2; do
  run "quiet pipelined ${W}x${H}" "--width $W --height $H"
  run "quiet no-pipeline ${W}x${H}" "--width $W --height $H --no-pipeline"
done
PIDS=""
for i in 1 2 3 4 5 6 7 8; do python3 -c "
import time
t=time.time()
while time.time()-t<240: pass" & PIDS="$PIDS $!"; done
sleep 1
for rep in 1 2; do
  run "8-busy-threads pipelined ${W}x${H}" "--width $W --height $H"
  run "8-busy-threads no-pipeline ${W}x${H}" "--width $W --height $H --no-pipeline"
done
kill $PIDS 2>/dev/null
wait 2>/dev/null
