#!/bin/bash
# round 6, call 26: in-step tile / staging sweep of the RGB chain's launch-per-conv shapes
mkdir -p gpurun_out/r6y
HCM_DEV_LIB=1 timeout 2400 python tools/shape_sweep_instep.py --out gpurun_out/r6y/sweep.json \
  --shapes 65536,512,256 65536,256,512 16384,1024,512 16384,256,2304 16384,1024,256 16384,512,1024 4096,512,4608 4096,2048,1024 4096,2048,512 4096,512,2048 \
  --choices 6 12 24 30 36 42 48 54 60 29 35 59 25 100 101 > gpurun_out/r6y/sweep.txt 2>&1
cat gpurun_out/r6y/sweep.txt
