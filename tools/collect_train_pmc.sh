#!/bin/bash
# rocprofv3 evidence of a steady-state TRAINING step (tools/train_bench.py, 16 queries x (1 panorama + 11 tiles), BF16X3):
# a kernel trace for the durations and separate --pmc passes (never with a trace domain other than the kernel trace).
# tools/summarize_train_pmc.py turns them into profiles/<tag>_pmc_train.json.
#   bash tools/collect_train_pmc.sh r05 [extra train_bench.py arguments]
tag=${1:-r06}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--batch 16 --streams 1 --steps 3 --warmup 2 $*"
cd /tmp && export TMPDIR=/tmp
run() {   # name, rocprofv3 options...
  d=$R/gpurun_out/${tag}_trainpmc_$1; shift
  rm -rf $d; mkdir -p $d
  rocprofv3 "$@" --output-format csv -d $d -o c -- python3 $R/tools/train_bench.py $ARGS > $d/stdout.txt 2>&1
}
run trace --kernel-trace
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA
run wait --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_MFMA
run lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES
cd $R
python3 tools/summarize_train_pmc.py $tag > gpurun_out/${tag}_pmc_train.json 2> gpurun_out/${tag}_pmc_train.err
find gpurun_out -name "*kernel_trace.csv" -size +40M -delete
find gpurun_out -name "*counter_collection.csv" -size +40M -delete
