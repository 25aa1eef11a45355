#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp5
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu > gpurun_out/exp5/tests.txt 2>&1
tail -5 gpurun_out/exp5/tests.txt
for pr in 1 0; do
  timeout 600 python bench.py --pair $pr --verbose --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp5/bench_pair$pr.json 2> gpurun_out/exp5/bench_pair$pr.err
done
timeout 600 python bench.py --pair 1 --qsplit 1 --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp5/bench_pair1_q1.json 2> gpurun_out/exp5/bench_pair1_q1.err
timeout 600 python bench.py --pair 1 --graph 0 --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp5/bench_pair1_g0.json 2> gpurun_out/exp5/bench_pair1_g0.err
head -c 600 gpurun_out/exp5/bench_pair1.json; echo; head -c 600 gpurun_out/exp5/bench_pair0.json
