#!/bin/bash
# Launch-by-launch listing of each encoder chain run ALONE (development library, HCM_SKIP mask) at batch B: gpurun_out/<tag>/seq_{rgb,depth,bert}_b<B>.txt
#   usage (through gpurun): bash tools/chain_listing.sh <tag> [B]
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/$1; B=${2:-64}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "12 rgb pack_frame" "11 depth avgpool2" "7 bert bert_embed"; do
  set -- $c
  rm -rf /tmp/cl_$2
  HCM_DEV_LIB=1 HCM_SKIP=$1 HCM_SERIAL=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/cl_$2 -o p --output-format csv -- python $REPO/tools/chain_step.py $B > /dev/null 2>&1
  python $REPO/tools/chain_seq.py $(find /tmp/cl_$2 -name "p_kernel_trace.csv" | head -1) $3 > $OUT/seq_$2_b$B.txt 2>&1
done
