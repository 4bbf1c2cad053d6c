#!/bin/bash
# round 6, call 35: GEMM vs reduction time of the split FC per slice count (kernel trace)
export HCM_DEV_LIB=1
REPO=$(pwd); mkdir -p $REPO/gpurun_out/r6ah; cd /tmp && export TMPDIR=/tmp
for S in 8 14 28 56; do
  HCM_SPLITK_FORCE=$S timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/sk$S -o p --output-format csv -- python $REPO/tools/splitk_bench.py 256 128 25088 > /dev/null 2>&1
  echo "S=$S"; python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/sk$S/**/p_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in d.items():
    if len(v) >= 100: print(f"  {k:70s} n={len(v)} avg {sum(v)/len(v):.1f} us min {min(v):.1f}")
PY
done 2>&1 | tee $REPO/gpurun_out/r6ah/trace.txt
