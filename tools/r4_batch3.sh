#!/bin/bash
# round-4 GPU batch 3: in-step vs alone per kernel (kernel traces of the same command, three-chain step vs HCM_SERIAL), step timeline
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r4b3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 8 --warmup 1 --prewarm 0 --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0"
STEPS=15       # 1 + 5 untimed + 1 warm-up + 8 timed
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt_conc -o p --output-format csv -- python $REPO/bench.py $ARGS > $OUT/kt_conc.log 2>&1
HCM_SERIAL=1 HCM_GRAPH=0 timeout 600 rocprofv3 --kernel-trace -d $OUT/kt_ser -o p --output-format csv -- python $REPO/bench.py $ARGS --no-graph > $OUT/kt_ser.log 2>&1
A=$(find $OUT/kt_conc -name "p_kernel_trace.csv" | head -1); B=$(find $OUT/kt_ser -name "p_kernel_trace.csv" | head -1)
python $REPO/tools/step_compare.py $A $B $STEPS > $OUT/step_compare.md 2>&1
python $REPO/tools/ktrace_summary.py $A $STEPS > $OUT/kt_conc.md
python $REPO/tools/ktrace_summary.py $B $STEPS > $OUT/kt_ser.md
gzip -c $A > $OUT/kt_conc.csv.gz
rm -rf $OUT/kt_conc $OUT/kt_ser
ls -la $OUT
