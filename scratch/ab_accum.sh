#!/bin/bash
# kernel time of the accumulate / solve kernels in the headline run, per library variant (rocprofv3 kernel stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  export VIPE_AMD_LIB=$R/scratch/lib/libvipe_$t.so
  rm -rf /tmp/acc_$t
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/acc_$t -o d -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-hipgraph --serial-operator > /tmp/acc_$t.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/acc_$t/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(x in r['Name'] for x in ('ba_accum','solve_band','ba_retract')):
        print('$t', r['Name'][:50], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
done
