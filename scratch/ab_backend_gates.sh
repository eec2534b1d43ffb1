#!/bin/bash
# the global BA's shadow: next pass's hidden-state gate part staged under it (update_batch) - whole-clip frames/s with / without
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do for f in 1 0; do
  VIPE_AMD_BACKEND_GATE_OVERLAP=$f python3 bench.py --mode video --frames 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['rank0_clip']; print('backend gate overlap $f: pass1', round(d['pass1'],1), 'through_global_ba', round(d['through_global_ba'],1), 'whole', round(d['slam_system_run'],1))"
done; done
