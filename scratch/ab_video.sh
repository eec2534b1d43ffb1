#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in "$@"; do
  export VIPE_AMD_LIB=$PWD/scratch/lib/libvipe_$t.so
  echo "== $t tests"; python3 -m pytest tests/test_gpu_parity.py -x -q -k "dense_ba or factor_graph or frontend or band or rig or slam_system" 2>&1 | tail -1
done
for rep in 1 2 3; do for t in "$@"; do
  export VIPE_AMD_LIB=$PWD/scratch/lib/libvipe_$t.so
  python3 bench.py --mode video --frames 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t video 512x384', round(d['value'],1), 'frames/s')"
done; done
for rep in 1 2; do for t in "$@"; do
  export VIPE_AMD_LIB=$PWD/scratch/lib/libvipe_$t.so
  python3 bench.py --mode video --frames 200 --height 328 --width 584 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t video 584x328', round(d['value'],1), 'frames/s')"
done; done
