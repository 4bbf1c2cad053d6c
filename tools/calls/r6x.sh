#!/bin/bash
# round 6, call 25: the RGB chain held back until BERT is k layers in (development build, HCM_RGB_GATE=k)
mkdir -p gpurun_out/r6x
for k in 1 2 3 4 6; do
  bash tools/ab.sh r6x/gate_$k "HCM_RGB_GATE=$k" 2 2>&1 | tail -2
  tail -2 gpurun_out/r6x/gate_$k/alt_1.err
done
