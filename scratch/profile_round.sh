#!/bin/bash
# Collect the committed profile set on the GPU box: kernel stats of the default bench command + two PMC passes.
# Usage (through gpurun): bash scratch/profile_round.sh <tag>
set -e
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
[ -n "$SKIP_STATS" ] || rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python $R/bench.py > $R/gpurun_out/${tag}_bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_fetch -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --prof-steps 0 > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_write -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --prof-steps 0 > $R/gpurun_out/${tag}_pmc_write.log 2>&1
echo done
