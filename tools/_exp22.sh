#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/exp22_tests.txt 2>&1; tail -8 gpurun_out/exp22_tests.txt
