mkdir -p gpurun_out/r6p
export HCM_DEV_LIB=1
A="--steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --no-kernel-probe --configs-leg 0 --gather-leg 0"
for rep in 1 2; do
  for g in 0 1 2 3 4 12 14 16; do
    v=$(HCM_DEPTH_GATE=$g timeout 300 python bench.py $A 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'])")
    echo "gate=$g rep=$rep: $v" >> gpurun_out/r6p/gate.txt
  done
done
cat gpurun_out/r6p/gate.txt
