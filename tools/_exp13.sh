#!/bin/bash
cd $GRAFT_REPO_ROOT
export AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/libagplace_hip_census.so
export AGP_KXR2_STAGGER=0
for L in layer1 layer2 layer3; do timeout 200 python tools/census2.py 64 $L 1 2>&1 | grep -v amdgpu.ids; done
timeout 200 python tools/census2.py 64 layer1 0 2>&1 | grep -v amdgpu.ids
