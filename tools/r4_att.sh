#!/bin/bash
set -u
REPO=$(pwd); cd $REPO
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "golden or ragged or varlen or bf16" 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --latency-leg 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['bert_gemms']['ffn1']['us_per_launch'])"; done
timeout 300 python bench.py --batch 1 --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --latency-leg 0 --no-kernel-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1', d['value'])"
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/att -o p --output-format csv -- python $REPO/bench.py --steps 4 --warmup 1 --prewarm 0 --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 > /dev/null 2>&1; python $REPO/tools/ktrace_summary.py $(find /tmp/att -name "p_kernel_trace.csv" | head -1) 11 | grep -i "attention\|layernorm"
