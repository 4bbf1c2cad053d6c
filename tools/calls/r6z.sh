#!/bin/bash
# round 6, call 27: wall time of the default bench.py run and of smoke() on a fresh box (final tree)
mkdir -p gpurun_out/r6z
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r6z/smoke.log 2>&1
tail -6 gpurun_out/r6z/smoke.log
( time python bench.py > gpurun_out/r6z/bench_default.json 2> gpurun_out/r6z/bench_default.err ) 2> gpurun_out/r6z/bench_time.txt
cat gpurun_out/r6z/bench_time.txt
python -c "
import json; d=json.load(open('gpurun_out/r6z/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['configs'])"
