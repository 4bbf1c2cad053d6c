#!/bin/bash
# Profile `bench.py` on the GPU box and write the summaries the judged numbers come from under profiles/ (via gpurun_out/).
#   usage (through gpurun):  bash tools/profile_bench.sh r2 [config]
# Passes (each its own process, counters never combined with API traces):
#   1. rocprofv3 --kernel-trace --stats          -> <tag>_kernel_trace_bench[_cfgN].md   (per-kernel average durations)
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE  \  -> <tag>_pmc_traffic_bench.md + pmc_traffic.json (HBM bytes, x2 gfx950 correction)
#   3. rocprofv3 --kernel-trace --pmc WRITE_SIZE  /
#   4. rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -> <tag>_pmc_mfma_bench.md
set -u
TAG=${1:-r2}
CFG=${2:-1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_cfg${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config $CFG --steps 4 --warmup 1 --prewarm 0 --sustain 0 --no-cpu-baseline --no-kernel-probe --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --configs-leg 0 --gather-leg 0 --host-procs-leg 0"
STEPS=11       # 1 + 5 untimed + 1 warm-up + 4 timed
SFX=""; [ "$CFG" != "1" ] && SFX="_cfg${CFG}"
run() { name=$1; shift; timeout 600 rocprofv3 "$@" -d $OUT/$name -o p --output-format csv -- python $REPO/bench.py $ARGS > $OUT/$name.log 2>&1; }
run kt --kernel-trace --stats
python $REPO/tools/ktrace_summary.py $(find $OUT/kt -name "p_kernel_trace.csv" | head -1) $STEPS > $OUT/${TAG}_kernel_trace_bench${SFX}.md
if [ "${3:-all}" = "all" ]; then
  run fetch --kernel-trace --pmc FETCH_SIZE
  run write --kernel-trace --pmc WRITE_SIZE
  python $REPO/tools/pmc_traffic.py $(find $OUT/fetch -name "p_counter_collection.csv" | head -1) $(find $OUT/write -name "p_counter_collection.csv" | head -1) $STEPS \
      $OUT/pmc_traffic${SFX}.json > $OUT/${TAG}_pmc_traffic_bench${SFX}.md
  run mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  python $REPO/tools/pmc_mfma.py $(find $OUT/mfma -name "p_counter_collection.csv" | head -1) $STEPS > $OUT/${TAG}_pmc_mfma_bench${SFX}.md
  if [ "$CFG" = "1" ]; then
    # the tool's own calibration: the kernel of nothing but MFMAs under the SAME counters, read by the SAME code (must say 95 %)
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/calib -o p --output-format csv -- python $REPO/tools/mfma_calib.py 2000 > $OUT/calib.log 2>&1
    python $REPO/tools/pmc_mfma.py $(find $OUT/calib -name "p_counter_collection.csv" | head -1) 1 --selfcheck > $OUT/${TAG}_pmc_mfma_selfcheck.md 2>&1
    rm -rf $OUT/calib
  fi
fi
# keep the merged-back payload small: summaries only
rm -rf $OUT/kt $OUT/fetch $OUT/write $OUT/mfma
ls -la $OUT
