#!/bin/bash
# Engine clock and socket power while bench.py runs (rocm-smi every 0.3 s beside a 12 s sustained run).
# usage (GPU box): bash tools/power_clock_sample.sh [bench args...]  -> gpurun_out/power_clock.log
mkdir -p gpurun_out
(python bench.py --no-cpu-baseline --no-kernel-probe --sustain 12 "$@" > gpurun_out/power_clock_bench.json 2>/dev/null &)
sleep 2
for i in $(seq 1 44); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e "s/.*: //" -e "s/=//g" | tr "\n" " "
    echo
    sleep 0.3
done > gpurun_out/power_clock.log
sleep 3
cat gpurun_out/power_clock.log
