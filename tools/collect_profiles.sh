#!/bin/bash
# Collect the rocprofv3 evidence that tools/summarize_profiles.py turns into profiles/<tag>_*.
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh r01
# (the program itself follows `--`; counters in their own passes, never with a trace domain other than the kernel trace)
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--no-cpu-baseline --no-knn --train-steps 0 --graph 0 --streams 1 --qsplit 1 --steps 5 --warmup 2"
mkdir -p $R/gpurun_out/${tag}_trace $R/gpurun_out/${tag}_pmc_fetch $R/gpurun_out/${tag}_pmc_write $R/gpurun_out/${tag}_pmc_mfma
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -o t -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_fetch -o c -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_write -o c -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/${tag}_pmc_mfma -o c -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cd $R && python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_err.log
tail -c 1500 gpurun_out/${tag}_bench_line.json
