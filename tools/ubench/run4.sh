for rep in 1 2 3; do for v in A B; do echo "lib$v: "; AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/variants/lib$v.so timeout 200 python tools/stem_bench.py 64 2>&1 | grep STEM; done; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "stem" 2>&1 | tail -3
