#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4fc; mkdir -p $OUT; cd $REPO
timeout 1500 python -m pytest tests/test_fusion_toggles_gpu.py tests/test_ops_gpu.py -m gpu -x -q > $OUT/test.log 2>&1; tail -3 $OUT/test.log
bash tools/r4_ab.sh r4fc/ilv HCM_DEEP_ILV3=0 3
HCM_DEV_LIB=1 timeout 600 python tools/step_determinism.py 3 2000 2>&1 | tail -1
R4_MODES=chain,forked,eager HCM_DEV_LIB=1 timeout 600 python tools/step_determinism.py 2 1000 2>&1 | tail -1
