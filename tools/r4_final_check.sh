#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4fc; mkdir -p $OUT; cd $REPO
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_integration_gpu.py tests/test_properties_gpu.py -m gpu -x -q > $OUT/test.log 2>&1; tail -3 $OUT/test.log
for b in 1 2; do timeout 300 python tools/act_host_profile.py $b 2>&1 | grep "^B="; done
HCM_DEV_LIB=1 HCM_IGEMM_TIME=1 HCM_GRAPH=0 HCM_SERIAL=1 timeout 600 python tools/shape_times.py 1 1 2>&1 >/dev/null | grep "M=80 "
