#!/bin/bash
# gpurun_out/prof (scratch/collect_profiles.sh a + b) -> the round's files under profiles/
set -e
cd /root/repo
R=${1:-r04}; O=gpurun_out/prof
cp $O/bench/b_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $O/bench41/b_kernel_stats.csv profiles/${R}_bench_41x73_kernel_stats.csv
cp $O/bench2/b_kernel_stats.csv profiles/${R}_bench_two_stream_kernel_stats.csv
cp $O/fetch/f_counter_collection.csv profiles/${R}_pmc_fetch_size.csv
cp $O/write/w_counter_collection.csv profiles/${R}_pmc_write_size.csv
cp $O/fetch41/f_counter_collection.csv profiles/${R}_pmc_fetch_size_41x73.csv
cp $O/write41/w_counter_collection.csv profiles/${R}_pmc_write_size_41x73.csv
cp $O/mfma_util.json profiles/${R}_mfma_util.json
cp $O/mfma41_util.json profiles/${R}_mfma_util_41x73.json
python3 profiles/summarize.py profiles/${R}_bench_kernel_stats.csv profiles/${R}_pmc_fetch_size.csv profiles/${R}_pmc_write_size.csv profiles/${R}_summary.json 39 > /dev/null
python3 profiles/summarize.py profiles/${R}_bench_41x73_kernel_stats.csv profiles/${R}_pmc_fetch_size_41x73.csv profiles/${R}_pmc_write_size_41x73.csv profiles/${R}_summary_41x73.json 39 > /dev/null
cp $O/video/v_kernel_stats.csv profiles/${R}_video_kernel_stats.csv
cp $O/video41/v_kernel_stats.csv profiles/${R}_video_584x328_kernel_stats.csv
cp $O/videok/v_kernel_stats.csv profiles/${R}_video_keep_every_4_kernel_stats.csv
cp $O/backend/v_kernel_stats.csv profiles/${R}_backend_kernel_stats.csv
cp $O/backend_alt/v_kernel_stats.csv profiles/${R}_backend_altcorr_kernel_stats.csv
tail -1 $O/bench_line.json > profiles/${R}_bench_line.json
ls -la profiles/${R}_* | wc -l
