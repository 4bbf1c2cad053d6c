#!/bin/bash
# round 6, call 34: what a K-slice count costs for SimpleCNN's FC (M = 256, N = 128, K = 25088), GEMM + reduction
export HCM_DEV_LIB=1
for S in 1 2 4 7 8 14 28 56; do HCM_SPLITK_FORCE=$S timeout 100 python tools/splitk_bench.py 256 128 25088 2>&1 | grep -v amdgpu; done
HCM_SPLITK_POW2=1 timeout 100 python tools/splitk_bench.py 256 128 25088 2>&1 | grep -v amdgpu
timeout 100 python tools/splitk_bench.py 256 128 25088 2>&1 | grep -v amdgpu
HCM_IGEMM_LOG=1 HCM_SPLITK_FORCE=56 timeout 100 python tools/splitk_bench.py 256 128 25088 2>&1 | grep "\[igemm\]"
HCM_IGEMM_LOG=1 HCM_SPLITK_FORCE=8 timeout 100 python tools/splitk_bench.py 256 128 25088 2>&1 | grep "\[igemm\]"
