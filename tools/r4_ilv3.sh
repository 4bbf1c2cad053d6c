#!/bin/bash
# ILV=3 (pipelined deep-ring loop): bit-identity, per-shape times, in-step A/B at B = 64 / 8 / 1
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ilv3; mkdir -p $OUT; cd $REPO
timeout 900 python -m pytest tests/test_fusion_toggles_gpu.py -m gpu -x -q -k deep_ring > $OUT/test.log 2>&1; tail -3 $OUT/test.log
HCM_DEV_LIB=1 HCM_IGEMM_TIME=1 HCM_GRAPH=0 HCM_SERIAL=1 timeout 600 python tools/shape_times.py 64 1 2> $OUT/shapes_base.md > /dev/null
HCM_DEEP_ILV3=1 HCM_DEV_LIB=1 HCM_IGEMM_TIME=1 HCM_GRAPH=0 HCM_SERIAL=1 timeout 600 python tools/shape_times.py 64 1 2> $OUT/shapes_ilv3.md > /dev/null
bash tools/r4_ab.sh r4ilv3/b64 HCM_DEEP_ILV3=1 3
bash tools/r4_ab.sh r4ilv3/b8 HCM_DEEP_ILV3=1 2 --batch 8
bash tools/r4_ab.sh r4ilv3/b1 HCM_DEEP_ILV3=1 2 --batch 1
