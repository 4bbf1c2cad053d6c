#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4b5; mkdir -p $OUT; cd $REPO
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "bottleneck_tail_next" > $OUT/pytest_bneck.txt 2>&1
export HCM_DEV_LIB=1
timeout 300 python tools/bneck_bench.py 128 32 128 1 > $OUT/bneck128_new.txt 2>&1
HCM_BNECK128_BM64=1 timeout 300 python tools/bneck_bench.py 128 32 128 1 > $OUT/bneck128_bm64.txt 2>&1
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_base_$i.json 2> $OUT/bench_base_$i.err
  HCM_BNECK128_BM64=1 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_bm64_$i.json 2> $OUT/bench_bm64_$i.err
  HCM_GEMM256_8PHASE=1 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --no-kernel-probe > $OUT/bench_8ph_$i.json 2> $OUT/bench_8ph_$i.err
done
