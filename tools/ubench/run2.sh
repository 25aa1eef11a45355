timeout 200 python tools/bblock_bench.py --reps 40 --rounds 3 --pool 1 2>&1 | grep -E "fused|bitwise"
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_basicblock" 2>&1 | tail -3
