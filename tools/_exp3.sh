#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/exp3
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/exp3/counters.txt 2>&1
for L in layer1 layer3; do
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/exp3/pmcA_$L -o c -- python3 $R/tools/conv_bench.py --prec 4 --batch 64 --res 1 --only $L --reps 5 > $R/gpurun_out/exp3/pmcA_$L.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/exp3/pmcB_$L -o c -- python3 $R/tools/conv_bench.py --prec 4 --batch 64 --res 1 --only $L --reps 5 > $R/gpurun_out/exp3/pmcB_$L.log 2>&1
done
ls -R $R/gpurun_out/exp3 | head -40
