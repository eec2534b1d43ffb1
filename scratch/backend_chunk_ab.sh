#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for ce in 1024 1792; do
VIPE_AMD_BACKEND_CHUNK_EDGES=$ce python3 bench.py --mode video --frames 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']['rank0_clip']
print('chunk $ce:', round(d['value'],1), 'frames/s; backend s', round(c['backend_seconds'],3), 'edges', c['backend_edges'])"
done; done
