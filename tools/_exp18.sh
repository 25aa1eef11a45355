#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/_h2d_probe.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
