#!/bin/bash
# round 6, call 28: determinism soak of the final tree (every step twice from the same inputs, forked-graph and eager replays against each other)
mkdir -p gpurun_out/r6aa
R4_FULL=1 R4_MODES=forked,eager timeout 900 python tools/step_determinism.py 64 1500 > gpurun_out/r6aa/full_b64.txt 2>&1; tail -3 gpurun_out/r6aa/full_b64.txt
R4_FULL=1 R4_MODES=forked,chain timeout 600 python tools/step_determinism.py 1 3000 > gpurun_out/r6aa/full_b1.txt 2>&1; tail -3 gpurun_out/r6aa/full_b1.txt
for hw in 128 192 384; do
  R4_DEPTH_HW=$hw timeout 600 python tools/step_determinism.py 3 2000 > gpurun_out/r6aa/small_hw$hw.txt 2>&1; tail -2 gpurun_out/r6aa/small_hw$hw.txt
done
for i in 1 2 3; do timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -k "config4" 2>&1 | tail -1; done > gpurun_out/r6aa/config4_x3.txt; cat gpurun_out/r6aa/config4_x3.txt
