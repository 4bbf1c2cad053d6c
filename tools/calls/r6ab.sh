#!/bin/bash
# round 6, call 29: rocprof summaries of configs 0 / 3 / 4 on the final tree
for c in 0 3 4; do bash tools/profile_bench.sh r6 $c > gpurun_out/r6ab_profile_cfg$c.log 2>&1; tail -3 gpurun_out/r6ab_profile_cfg$c.log; done
head -8 gpurun_out/prof_r6_cfg4/r6_kernel_trace_bench_cfg4.md
