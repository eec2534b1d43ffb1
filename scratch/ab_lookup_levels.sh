#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 scratch/lookup_levels.py 2>/dev/null | tail -1
for l in 0 1 2 3; do VIPE_AMD_LIB=$R/scratch/lib/libvipe_skipl$l.so python3 scratch/lookup_levels.py 2>/dev/null | tail -1; done
