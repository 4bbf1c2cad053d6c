#!/bin/bash
# Evidence set of the tree as it is: profiles (configs 1, 0, 3, 4), per-shape table, in-step chain timeline, chain listings, bench lines.
#   usage (through gpurun): bash tools/evidence.sh <tag, e.g. r5>
# Everything lands in gpurun_out/<tag>ev/ ; the summaries are copied into profiles/ afterwards (tools/README.md says which file comes from which tool).
set -u
TAG=${1:-r5}
REPO=$(pwd); OUT=$REPO/gpurun_out/${TAG}ev; mkdir -p $OUT; cd $REPO
for c in 1 0 3 4; do bash tools/profile_bench.sh $TAG $c > $OUT/profile_cfg$c.log 2>&1; done
cp gpurun_out/prof_${TAG}_cfg*/${TAG}_*.md gpurun_out/prof_${TAG}_cfg*/pmc_traffic*.json $OUT/ 2>/dev/null
bash tools/chain_listing.sh ${TAG}ev 64 > /dev/null 2>&1
timeout 400 python tools/chain_times.py 64 > $OUT/chain_times_b64.txt 2>&1
cd $REPO
HCM_DEV_LIB=1 HCM_IGEMM_TIME=1 HCM_GRAPH=0 HCM_SERIAL=1 timeout 600 python tools/shape_times.py 64 1 2> $OUT/${TAG}_igemm_shapes_raw.md > /dev/null
HCM_DEV_LIB=1 timeout 600 python tools/step_marks.py 64 > $OUT/marks_b64.txt 2>&1
HCM_DEV_LIB=1 timeout 600 python tools/step_marks.py 1 > $OUT/marks_b1.txt 2>&1
for b in 1 2 4; do timeout 600 python tools/act_host_profile.py $b 2>&1 | grep "^B="; done > $OUT/host_b1.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in 0 3 4; do timeout 600 python bench.py --config $c --sustain 0 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; done
for b in 1 4 8 16 32 128 256; do timeout 600 python bench.py --batch $b --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --configs-leg 0 --gather-leg 0 --host-procs-leg 0 > $OUT/bench_b$b.json 2> /dev/null; done
timeout 600 python bench.py --h2d --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --configs-leg 0 --gather-leg 0 --host-procs-leg 0 > $OUT/bench_h2d.json 2> /dev/null
timeout 600 python bench.py --reuse-instruction --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --configs-leg 0 --gather-leg 0 --host-procs-leg 0 > $OUT/bench_reuse.json 2> /dev/null
timeout 600 python bench.py --precision fp32 --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --configs-leg 0 --gather-leg 0 --host-procs-leg 0 > $OUT/bench_fp32.json 2> /dev/null
timeout 1500 python bench.py --cpu-batches --sustain 0 --no-kernel-probe --bf16-leg 0 --latency-leg 0 --configs-leg 0 --gather-leg 0 --host-procs-leg 0 > $OUT/bench_cpu_batches.json 2> /dev/null
ls $OUT
