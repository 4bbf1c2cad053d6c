#!/bin/bash
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r4b4
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_fusion_toggles_gpu.py -x -q -k "fused_rgb_trunk" > $OUT/pytest_toggles.txt 2>&1
timeout 600 python -m pytest tests/test_integration_gpu.py -x -q -k "library_all_gather" > $OUT/pytest_gather.txt 2>&1
export HCM_DEV_LIB=1
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_base_$i.json 2> $OUT/bench_base_$i.err
  HCM_NO_BNECK256=1 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_no256_$i.json 2> $OUT/bench_no256_$i.err
  HCM_NO_BNECK_XCD=1 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_noxcd_$i.json 2> $OUT/bench_noxcd_$i.err
  HCM_GEMM256_FREE=3 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_free3_$i.json 2> $OUT/bench_free3_$i.err
done
ls $OUT
