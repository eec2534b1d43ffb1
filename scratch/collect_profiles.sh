#!/bin/bash
# run on the GPU box (gpurun): the round's judged profiles into gpurun_out/prof/ (copied to profiles/ afterwards)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-hipgraph --serial-operator > $O/bench.log 2>&1
echo bench stats done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench2 -o b -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-hipgraph > $O/bench2.log 2>&1
echo bench two-stream stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-hipgraph --serial-operator > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-hipgraph --serial-operator > $O/write.log 2>&1
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/mfma -o m -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-hipgraph --serial-operator > $O/mfma.log 2>&1
python3 $R/profiles/mfma_util.py $(find $O/mfma -name "*counter_collection.csv" | head -1) $O/mfma_util.json > /dev/null
echo mfma utilisation done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/video -o v -- python3 $R/bench.py --mode video --frames 200 > $O/video.log 2>&1
echo video done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/videob -o v -- python3 $R/bench.py --mode video --frames 200 --with-backend > $O/videob.log 2>&1
echo video+backend done
python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
echo bench line done
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*.db" -delete
du -sh $O
find $O -name "*.csv" | head -40
