#!/bin/bash
# kNN leg (100k x 256, 4096 queries, k = 20): kernel trace + PMC passes (counters in their own runs, kernel trace only).
# bash tools/collect_knn_pmc.sh r03  ->  gpurun_out/<tag>_knn*  ; python tools/summarize_knn_pmc.py r03 -> profiles/<tag>_pmc_knn.json
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
CMD="python3 $R/tools/knn_bench.py --prec 4 --reps 20"
mkdir -p $R/gpurun_out/${tag}_knn $R/gpurun_out/${tag}_knn_fetch $R/gpurun_out/${tag}_knn_write $R/gpurun_out/${tag}_knn_sq
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_knn -o k -- $CMD > $R/gpurun_out/${tag}_knn/stdout.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_knn_fetch -o c -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_knn_write -o c -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${tag}_knn_sq -o c -- $CMD > /dev/null 2>&1
find $R/gpurun_out -name "*kernel_trace.csv" -size +40M -delete
cat $R/gpurun_out/${tag}_knn/stdout.txt | tail -2
