#!/bin/bash
set -u
REPO=$(pwd); cd $REPO
for b in 24 32 48 128; do for cg in 0 1; do
  timeout 300 python bench.py --h2d --batch $b --chain-graphs $cg --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h2d B=$b chain=$cg', d['value'], d['ms_per_step'], d['host_us_per_step'])"
done; done
