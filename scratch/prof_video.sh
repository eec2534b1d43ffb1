set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof3a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/video -o v -- python3 $R/bench.py --mode video --frames 200 > $O/video.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*.db" -delete
tail -2 $O/video.log
