#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -x -s -k "realistic_cloud" 2>&1 | grep "FULLCLOUD\|passed\|failed\|Error" | head
