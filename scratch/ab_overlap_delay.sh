#!/bin/bash
# head start of the BA's solve kernel before a staged-gate piece is released (vipe_overlap_fn, csrc/ba.hip overlap_piece):
# headline it/s with the 4 us delay kernel (default), without it, and with 1 / 16 us
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do for t in 400 0 100 1600; do
  VIPE_AMD_OVERLAP_DELAY_TICKS=$t python3 bench.py --no-secondary --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('delay ticks $t:', round(d['value'],1), 'it/s')"
done; done
