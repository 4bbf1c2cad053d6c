#!/bin/bash
# round 6, call 23: the rocprof summaries of the default bench configuration on the FINAL tree (kernel trace, HBM traffic, MFMA busy)
bash tools/profile_bench.sh r6 1 > gpurun_out/r6v_profile.log 2>&1
tail -5 gpurun_out/r6v_profile.log
head -30 gpurun_out/prof_r6_cfg1/r6_kernel_trace_bench.md
