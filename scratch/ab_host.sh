#!/bin/bash
# alternate two versions of host files (scratch/abvar/old, scratch/abvar/new -> vipe_amd/slam/) on one box: video frames/s
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for v in old new; do
  cp scratch/abvar/$v/*.py vipe_amd/slam/
  python3 bench.py --mode video --frames 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v video 512x384', round(d['value'],1), 'frames/s')"
done; done
cp scratch/abvar/new/*.py vipe_amd/slam/
