#!/bin/bash
# round 6, call 32: determinism soak again, with the register-weights cross-modal layer in the step
mkdir -p gpurun_out/r6ae
R4_FULL=1 R4_MODES=forked,eager timeout 900 python tools/step_determinism.py 64 1000 2>&1 | grep -v amdgpu | tail -1
R4_FULL=1 R4_MODES=forked,chain timeout 600 python tools/step_determinism.py 1 2000 2>&1 | grep -v amdgpu | tail -1
R4_DEPTH_HW=256 timeout 600 python tools/step_determinism.py 3 2000 2>&1 | grep -v amdgpu | tail -1
for i in 1 2; do timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -k "config4" 2>&1 | tail -1; done
