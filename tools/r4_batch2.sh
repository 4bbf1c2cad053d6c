#!/bin/bash
# round-4 GPU batch 2: fused RGB layer3 bottleneck (256 mid channels): parity + timing + in-step A/B; free-running GEMM variants 17-19
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r4b2
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "bottleneck_tail_next_fused" > $OUT/pytest_bneck.txt 2>&1
timeout 300 python tools/bneck256_bench.py 128 16 > $OUT/bneck256_bench.txt 2>&1
timeout 300 python tools/bneck256_bench.py 16 16 >> $OUT/bneck256_bench.txt 2>&1
export HCM_DEV_LIB=1
( VAR=17 timeout 300 python tools/gemm256_sched_check.py; VAR=18 RES=1 timeout 300 python tools/gemm256_sched_check.py ) > $OUT/sched_check.txt 2>&1
KS=768,3072 timeout 300 python tools/gemm256_ksweep.py 5120 3072 0 12,17,18,19 > $OUT/ksweep_act0.txt 2>&1
KS=768 timeout 300 python tools/gemm256_ksweep.py 5120 3072 2 12,17,18,19 > $OUT/ksweep_gelu.txt 2>&1
KS=768 timeout 300 python tools/gemm256_ksweep.py 5120 2304 0 12,17,18,19 > $OUT/ksweep_qkv.txt 2>&1
# in-step A/B (development library): layer3 fusion on/off; free-running GEMM forms
for i in 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 > $OUT/bench_base_$i.json 2> $OUT/bench_base_$i.err
  HCM_NO_BNECK256=1 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 > $OUT/bench_no256_$i.json 2> $OUT/bench_no256_$i.err
  HCM_GEMM256_FREE=2 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 > $OUT/bench_free2_$i.json 2> $OUT/bench_free2_$i.err
  HCM_GEMM256_FREE=3 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 > $OUT/bench_free3_$i.json 2> $OUT/bench_free3_$i.err
done
unset HCM_DEV_LIB
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q > $OUT/pytest_parity.txt 2>&1
ls $OUT
