#!/bin/bash
# round-4 evidence set of the FINAL state: profiles (configs 1, 0, 3, 4), per-shape table, in-step chain timeline, bench lines.  Everything lands in
# gpurun_out/r4ev/ ; the summaries are copied into profiles/ by hand afterwards.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ev; mkdir -p $OUT; cd $REPO
for c in 1 0 3 4; do bash tools/profile_bench.sh r4 $c > $OUT/profile_cfg$c.log 2>&1; done
cp gpurun_out/prof_r4_cfg*/r4_*.md gpurun_out/prof_r4_cfg*/pmc_traffic*.json $OUT/ 2>/dev/null
cd $REPO
HCM_DEV_LIB=1 HCM_IGEMM_TIME=1 HCM_GRAPH=0 HCM_SERIAL=1 timeout 600 python tools/shape_times.py 64 1 2> $OUT/r4_igemm_shapes_raw.md > /dev/null
HCM_DEV_LIB=1 timeout 600 python tools/step_marks.py 64 > $OUT/marks_b64.txt 2>&1
HCM_DEV_LIB=1 timeout 600 python tools/step_marks.py 1 > $OUT/marks_b1.txt 2>&1
for b in 1 2 4; do timeout 600 python tools/act_host_profile.py $b 2>&1 | grep "^B="; done > $OUT/host_b1.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in 0 3 4; do timeout 600 python bench.py --config $c --sustain 0 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; done
for b in 1 4 8 16 32 128 256; do timeout 600 python bench.py --batch $b --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 > $OUT/bench_b$b.json 2> /dev/null; done
timeout 600 python bench.py --h2d --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 > $OUT/bench_h2d.json 2> /dev/null
timeout 600 python bench.py --reuse-instruction --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 > $OUT/bench_reuse.json 2> /dev/null
timeout 600 python bench.py --precision fp32 --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 > $OUT/bench_fp32.json 2> /dev/null
timeout 1500 python bench.py --cpu-batches --sustain 0 --no-kernel-probe --bf16-leg 0 --latency-leg 0 > $OUT/bench_cpu_batches.json 2> /dev/null
ls $OUT
