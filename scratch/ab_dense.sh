#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  export VIPE_AMD_LIB=$R/scratch/lib/libvipe_$t.so
  rm -rf /tmp/dn_$t
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dn_$t -o d -- python3 $R/scratch/dense_solve_time.py > /tmp/dn_$t.log 2>&1)
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/dn_$t/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'solve_dense' in r['Name'] or 'ba_walk' in r['Name'] or 'ba_schur' in r['Name']:
        print('$t', r['Name'][:40], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
done
