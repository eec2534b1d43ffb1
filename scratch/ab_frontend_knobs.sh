#!/bin/bash
# frontend (pass 1) frames/s of SLAMSystem.run against the thresholds from which the operator's second stream and the
# staged gate state under the BA are used (defaults 64 / 64 active edges; the frontend's windows hold 40-48)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for cfg in "64 64" "32 64" "1 64" "64 32" "32 32"; do
  set -- $cfg
  VIPE_AMD_OP_SIDE_MIN_EDGES=$1 VIPE_AMD_GATE_OVERLAP_MIN_EDGES=$2 python3 bench.py --mode video --frames 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['rank0_clip']; print('op_side>=$1 gate_overlap>=$2: pass1', round(d['pass1'],1), 'whole', round(d['slam_system_run'],1))"
done; done
