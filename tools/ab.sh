#!/bin/bash
# same-box interleaved A/B of bench.py under the development library: usage ab.sh <outdir> <ENVVAR=1> [runs] [extra bench args]
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/$1; mkdir -p $OUT; cd $REPO
VAR=$2; N=${3:-3}; shift 3 || true
export HCM_DEV_LIB=1
A="--steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --no-kernel-probe --configs-leg 0 --gather-leg 0 --host-procs-leg 0"
for i in $(seq 1 $N); do
  timeout 300 python bench.py $A "$@" > $OUT/base_$i.json 2> $OUT/base_$i.err
  env $VAR timeout 300 python bench.py $A "$@" > $OUT/alt_$i.json 2> $OUT/alt_$i.err
done
python - <<PY
import json, glob
for k in ("base", "alt"):
    v = []
    for f in sorted(glob.glob("$OUT/%s_*.json" % k)):
        try: v.append(json.load(open(f))["value"])
        except Exception: v.append(None)
    print(k, v)
PY
