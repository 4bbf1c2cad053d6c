#!/bin/bash
# round 6, call 30: the cross-modal layer with its weights straight into registers (vla_post_wf_kernel): op parity + bit equality, toggle tests, timing
mkdir -p gpurun_out/r6ac
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "fused_cross_modal_layer_op" > gpurun_out/r6ac/op_test.log 2>&1; tail -3 gpurun_out/r6ac/op_test.log
timeout 900 python -m pytest tests/test_fusion_toggles_gpu.py -q -k "cross_modal" > gpurun_out/r6ac/toggle.log 2>&1; tail -3 gpurun_out/r6ac/toggle.log
for BL in "64 80" "256 80" "128 160" "1 80"; do timeout 120 python tools/vla_op_bench.py $BL 2>&1 | grep -v amdgpu; done > gpurun_out/r6ac/op_bench.txt; cat gpurun_out/r6ac/op_bench.txt
bash tools/ab.sh r6ac/ab_cfg1 "HCM_NO_VLA_WFRAG=1" 3 2>&1 | tail -2
bash tools/ab.sh r6ac/ab_cfg4 "HCM_NO_VLA_WFRAG=1" 2 --config 4 2>&1 | tail -2
for i in 1 2; do for f in 1 0; do
  timeout 200 python bench.py --config 3 --steps 2000 --warmup 50 --sustain 0 --no-cpu-baseline --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --no-kernel-probe --configs-leg 0 --gather-leg 0 --host-procs-leg 0 $( [ $f = 0 ] && echo --probe-lds-ring ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 frag=$f', d['value'], d['ms_per_step'])"
done; done
