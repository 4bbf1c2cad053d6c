#!/bin/bash
# round 6, call 24: FFN2 (M = 5120, N = 768, K = 3072) on the 256 x 128 tile inside the step -- fewer bytes through each CU per flop, half the CUs
mkdir -p gpurun_out/r6w
export HCM_DEV_LIB=1
A="--steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --no-kernel-probe --configs-leg 0 --gather-leg 0 --host-procs-leg 0"
HCM_IGEMM_LOG=1 timeout 300 python bench.py $A > gpurun_out/r6w/log_run.json 2> gpurun_out/r6w/shapes.log
grep "\[igemm\]" gpurun_out/r6w/shapes.log | sort -u > gpurun_out/r6w/shapes.txt; wc -l gpurun_out/r6w/shapes.txt
grep "M=5120" gpurun_out/r6w/shapes.txt
for c in 100 101 103 105; do
  bash tools/ab.sh r6w/ffn2_$c "HCM_IGEMM_SHAPE_FORCE=5120,768,3072:$c" 2 2>&1 | tail -2
done
