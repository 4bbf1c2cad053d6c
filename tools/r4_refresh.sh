#!/bin/bash
# final refresh on the final tree: configs[1] profile set + the default bench line
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ev; mkdir -p $OUT; cd $REPO
bash tools/profile_bench.sh r4 1 > $OUT/profile_cfg1.log 2>&1
cp gpurun_out/prof_r4_cfg1/r4_*.md gpurun_out/prof_r4_cfg1/pmc_traffic.json $OUT/ 2>/dev/null
cd $REPO
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.json
