#!/bin/bash
# A/B of the generic gather-GEMM's configurations on the voxel step (AGP_IGEMM_VARIANT)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for d in 0; do
  mkdir -p $R/gpurun_out/tsab
  AGP_IGEMM_VARIANT=$d rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tsab -o t -- python3 $R/bench.py --vox --no-cpu-baseline --no-knn --train-steps 0 --default-prec-leg 0 --graph 0 --streams 1 --qsplit 1 --steps 4 --warmup 2 > /dev/null 2>&1
  echo "variant $d"; python3 $R/tools/vox_timeline.py $(find $R/gpurun_out/tsab -name "*kernel_trace.csv" | head -1) | tail -1
  rm -rf $R/gpurun_out/tsab
done
