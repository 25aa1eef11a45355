#!/bin/bash
# A/B of the gather-GEMM on the voxel step: AGP_IGEMM_DBG=128 walks every tap (no per-tile tap list), AGP_IGEMM_VARIANT picks a tile shape
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for d in ${AB_LIST:-0 128}; do
  mkdir -p $R/gpurun_out/tsab
  AGP_IGEMM_DBG=$d rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tsab -o t -- python3 $R/bench.py --vox --no-cpu-baseline --no-knn --train-steps 0 --default-prec-leg 0 --graph 0 --streams 1 --qsplit 1 --steps 4 --warmup 2 > /dev/null 2>&1
  echo "AGP_IGEMM_DBG $d"; python3 $R/tools/vox_timeline.py $(find $R/gpurun_out/tsab -name "*kernel_trace.csv" | head -1) | tail -1
  rm -rf $R/gpurun_out/tsab
done
