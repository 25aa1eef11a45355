#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp1
for p in 4 2; do
  python bench.py --prec $p --verbose --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp1/bench_p$p.json 2> gpurun_out/exp1/bench_p$p.err
done
for p in 4 2; do for b in 64 32; do
  python tools/conv_bench.py --prec $p --batch $b --res 1 > gpurun_out/exp1/conv_p${p}_b$b.txt 2>&1
done; done
python tools/conv_bench.py --prec 4 --batch 64 --res 1 --zeros 1 > gpurun_out/exp1/conv_p4_b64_zeros.txt 2>&1
tail -n 3 gpurun_out/exp1/conv_p4_b64.txt
