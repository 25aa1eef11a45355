timeout 200 python tools/bblock_bench.py --reps 40 --rounds 3 --pool 0 2>&1 | grep -E "round|bitwise"
AGP_FB_M16=0 timeout 200 python tools/bblock_bench.py --reps 40 --rounds 2 --pool 0 2>&1 | grep -E "round 1|bitwise"
timeout 200 python tools/bblock_bench.py --reps 40 --rounds 2 --pool 1 2>&1 | grep -E "round 1|bitwise"
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_basicblock" 2>&1 | tail -3
