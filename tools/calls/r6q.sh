OUT=gpurun_out/r6ev; mkdir -p $OUT
for c in 0 3 4; do bash tools/profile_bench.sh r6 $c > $OUT/profile_cfg$c.log 2>&1; done
cp gpurun_out/prof_r6_cfg*/r6_*.md gpurun_out/prof_r6_cfg*/pmc_traffic*.json $OUT/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in 0 3 4; do timeout 600 python bench.py --config $c --sustain 0 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; done
X="--sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --configs-leg 0 --gather-leg 0"
for b in 1 16 128 256; do timeout 600 python bench.py --batch $b $X > $OUT/bench_b$b.json 2> /dev/null; done
timeout 600 python bench.py --precision fp32 $X > $OUT/bench_fp32.json 2> /dev/null
HCM_DEV_LIB=1 timeout 600 python tools/step_marks.py 1 > $OUT/marks_b1.txt 2>&1
ls $OUT
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6ev/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
