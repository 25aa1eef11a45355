#!/bin/bash
# Collect the rocprofv3 evidence that tools/summarize_profiles.py turns into profiles/<tag>_*.
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh r01
# (the program itself follows `--`; counters in their own passes, never with a trace domain other than the kernel trace)
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--no-cpu-baseline --no-knn --no-netvlad --vox-leg 0 --windows 1 --train-steps 0 --default-prec-leg 0 --graph 0 --streams 1 --qsplit 1 --steps 20 --warmup 2"
mkdir -p $R/gpurun_out/${tag}_trace $R/gpurun_out/${tag}_pmc_fetch $R/gpurun_out/${tag}_pmc_write $R/gpurun_out/${tag}_pmc_mfma
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -o t -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_fetch -o c -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_write -o c -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/${tag}_pmc_mfma -o c -- python3 $R/bench.py $ARGS > /dev/null 2>&1
# where the waves of the conv kernels wait: SQ_WAIT_ANY = parked on s_waitcnt / s_barrier, SQ_WAIT_INST_ANY = issue
# stalls, SQ_WAIT_INST_LDS = LDS issue stalls, SQ_ACTIVE_INST_ANY; all in quad-cycles like SQ_WAVE_CYCLES
mkdir -p $R/gpurun_out/${tag}_pmc_wait
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/${tag}_pmc_wait -o c -- python3 $R/bench.py $ARGS > /dev/null 2>&1
# kNN leg (100k x 256, k = 20) and one training step, kernel traces only
mkdir -p $R/gpurun_out/${tag}_train
bash $R/tools/collect_knn_pmc.sh $tag > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_train -o t -- python3 $R/tools/train_bench.py --batch 16 --streams 2 --steps 4 --warmup 2 > $R/gpurun_out/${tag}_train/stdout.txt 2>&1
python3 $R/tools/step_profile.py $(find $R/gpurun_out/${tag}_train -name "*kernel_trace.csv" | head -1) 2 > $R/gpurun_out/${tag}_train_step_kernels.txt 2>&1
find $R/gpurun_out -name "*kernel_trace.csv" -size +40M -delete
# the step with the sparse-voxel branch from coords (serial time per kernel class + the bench figure) and the training step with it
cd $R
bash tools/vox_profile.sh ${tag}_voxtrace > /dev/null 2>&1
bash tools/train_vox_profile.sh ${tag} > /dev/null 2>&1
python3 bench.py --vox --no-cpu-baseline --no-knn --train-steps 0 --default-prec-leg 0 > gpurun_out/${tag}_vox_line.json 2> /dev/null
# PMC passes of a steady-state training step (profiles/<tag>_pmc_train.json)
bash tools/collect_train_pmc.sh $tag > /dev/null 2>&1
# the summaries (into profiles/), THEN the bench line -- it quotes HBM traffic from them while their csrc_sha16 matches -- and a
# copy of everything under gpurun_out/ (the only directory that travels back from the GPU box)
python3 tools/summarize_profiles.py $tag 4 > /dev/null 2> gpurun_out/${tag}_summarize.err
python3 tools/summarize_knn_pmc.py $tag > /dev/null 2>> gpurun_out/${tag}_summarize.err
python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_err.log
cp gpurun_out/${tag}_bench_line.json profiles/${tag}_bench_line.json
mkdir -p gpurun_out/${tag}_profiles && cp profiles/${tag}_* gpurun_out/${tag}_profiles/ 2>/dev/null
tail -c 1500 gpurun_out/${tag}_bench_line.json
