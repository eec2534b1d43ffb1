#!/bin/bash
# run on the GPU box (gpurun): the round's judged profiles into gpurun_out/prof/ (copied to profiles/ afterwards)
# usage: collect_profiles.sh [part]   part = a (headline + 41x73 kernel stats, counters), b (video, backend altcorr, bench line)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PART=${1:-ab}
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --no-hipgraph --serial-operator"
G="--height 328 --width 584"
if [[ $PART == *a* ]]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- $B > $O/bench.log 2>&1
echo bench stats done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench41 -o b -- $B $G > $O/bench41.log 2>&1
echo bench 41x73 stats done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench2 -o b -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-hipgraph > $O/bench2.log 2>&1
echo bench two-stream stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $B > $O/write.log 2>&1
echo fetch / write done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch41 -o f -- $B $G > $O/fetch41.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write41 -o w -- $B $G > $O/write41.log 2>&1
echo fetch / write 41x73 done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/mfma -o m -- $B > $O/mfma.log 2>&1
python3 $R/profiles/mfma_util.py $(find $O/mfma -name "*counter_collection.csv" | head -1) $O/mfma_util.json > /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/mfma41 -o m -- $B $G > $O/mfma41.log 2>&1
python3 $R/profiles/mfma_util.py $(find $O/mfma41 -name "*counter_collection.csv" | head -1) $O/mfma41_util.json > /dev/null
echo mfma utilisation done
fi
if [[ $PART == *b* ]]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/video -o v -- python3 $R/bench.py --mode video --frames 200 > $O/video.log 2>&1
echo video done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/video41 -o v -- python3 $R/bench.py --mode video --frames 200 $G > $O/video41.log 2>&1
echo video 584x328 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/videok -o v -- python3 $R/bench.py --mode video --frames 400 --keep-every 4 > $O/videok.log 2>&1
echo video keep-every-4 done
export VIPE_AMD_BACKEND_ALTCORR=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/backend_alt -o v -- python3 $R/bench.py --mode backend --steps 5 --warmup 2 > $O/backend_alt.log 2>&1
unset VIPE_AMD_BACKEND_ALTCORR
echo backend altcorr done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/backend -o v -- python3 $R/bench.py --mode backend --steps 5 --warmup 2 > $O/backend.log 2>&1
echo backend done
python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
echo bench line done
fi
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*.db" -delete
du -sh $O
