#!/bin/bash
# per-kernel durations with every kernel alone on the chip (HCM_SERIAL=1, no graph) -> gpurun_out/$1/kt_ser.md
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/${1:-r4kt}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 8 --warmup 1 --prewarm 0 --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --no-graph"
HCM_SERIAL=1 HCM_GRAPH=0 timeout 600 rocprofv3 --kernel-trace -d $OUT/kt_ser -o p --output-format csv -- python $REPO/bench.py $ARGS > $OUT/kt_ser.log 2>&1
B=$(find $OUT/kt_ser -name "p_kernel_trace.csv" | head -1)
python $REPO/tools/ktrace_summary.py $B 15 > $OUT/kt_ser.md
rm -rf $OUT/kt_ser
