REPO=$GRAFT_REPO_ROOT; cd $REPO; mkdir -p gpurun_out/bf16prof
OUT=$REPO/gpurun_out/bf16prof
cd /tmp && export TMPDIR=/tmp
ARGS="--config 1 --steps 4 --warmup 1 --prewarm 0 --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 --h2d-leg 0"
for p in bf16 fp16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$p -o p --output-format csv -- python $REPO/bench.py $ARGS --precision $p > $OUT/kt_$p.log 2>&1
  python $REPO/tools/ktrace_summary.py $(find $OUT/kt_$p -name "p_kernel_trace.csv" | head -1) 11 > $OUT/kernel_trace_$p.md
  rm -rf $OUT/kt_$p
done
head -45 $OUT/kernel_trace_bf16.md | cut -c1-150
