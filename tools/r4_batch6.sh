#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4b6; mkdir -p $OUT; cd $REPO
export HCM_DEV_LIB=1
python tools/step_marks.py 1 2>&1 | grep -E "^##|depth.end|rgb.end|bert.end|tail.end" > $OUT/marks_b1_onload.txt
HCM_NO_GN_ONLOAD=1 python tools/step_marks.py 1 2>&1 | grep -E "^##|depth.end|rgb.end|bert.end|tail.end" > $OUT/marks_b1_apply.txt
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_base_$i.json 2> $OUT/bench_base_$i.err
  HCM_NO_GN_ONLOAD=1 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_apply_$i.json 2> $OUT/bench_apply_$i.err
done
for b in 1 8; do
  timeout 300 python bench.py --batch $b --steps 100 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_b${b}_base.json 2> /dev/null
  HCM_NO_GN_ONLOAD=1 timeout 300 python bench.py --batch $b --steps 100 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_b${b}_apply.json 2> /dev/null
done
