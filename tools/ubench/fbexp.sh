for d in 0 1; do echo "== AGP_FB_DBG=$d"; AGP_FB_DBG=$d timeout 120 python tools/bblock_bench.py --reps 40 --rounds 3 2>&1 | grep -E "fused|bitwise"; done
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_basicblock" 2>&1 | tail -3
