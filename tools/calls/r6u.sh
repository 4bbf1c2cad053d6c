#!/bin/bash
# round 6, call 22: the SimpleCNN overflow calibration test + the whole GPU suite on the final tree
mkdir -p gpurun_out/r6u
timeout 600 python -m pytest tests/test_trained_like_weights_gpu.py -q -s -k simplecnn > gpurun_out/r6u/simplecnn_overflow.log 2>&1
tail -15 gpurun_out/r6u/simplecnn_overflow.log
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r6u/gpu_suite.log 2>&1
tail -5 gpurun_out/r6u/gpu_suite.log
