#!/bin/bash
# round 6, call 33: split-K with one odd factor (SimpleCNN's 25088-wide FC: 8 -> 56 slices): op test, the SimpleCNN model / probe tests, configs[3] A/B
mkdir -p gpurun_out/r6af
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "long_k_split or test_linear" > gpurun_out/r6af/op.log 2>&1; tail -3 gpurun_out/r6af/op.log
timeout 1500 python -m pytest tests -q -m gpu -k "simplecnn or SimpleCNN or config3 or probe or configs3 or low_level" > gpurun_out/r6af/models.log 2>&1; tail -3 gpurun_out/r6af/models.log
A="--config 3 --steps 2000 --warmup 50 --sustain 0 --no-cpu-baseline --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --no-kernel-probe --configs-leg 0 --gather-leg 0 --host-procs-leg 0"
for i in 1 2 3; do
  HCM_DEV_LIB=1 timeout 200 python bench.py $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 odd-factor', d['value'], d['ms_per_step'])"
  HCM_DEV_LIB=1 HCM_SPLITK_POW2=1 timeout 200 python bench.py $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 pow2 only ', d['value'], d['ms_per_step'])"
done
