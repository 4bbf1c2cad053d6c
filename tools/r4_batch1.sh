#!/bin/bash
# round-4 GPU batch 1: free-running 256x256 GEMM (variant 12) vs the 8-phase form, phase stamps, matrix-pipe counter calibration
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r4b1
mkdir -p $OUT
export HCM_DEV_LIB=1
cd $REPO
( VAR=12 timeout 300 python tools/gemm256_sched_check.py; VAR=12 RES=1 timeout 300 python tools/gemm256_sched_check.py ) > $OUT/sched_check.txt 2>&1
KS=768,1536,3072,6144 timeout 300 python tools/gemm256_ksweep.py 5120 3072 0 12 > $OUT/ksweep_act0.txt 2>&1
KS=768 timeout 300 python tools/gemm256_ksweep.py 5120 3072 2 12 > $OUT/ksweep_gelu.txt 2>&1
KS=768 timeout 300 python tools/gemm256_ksweep.py 5120 2304 0 12 > $OUT/ksweep_qkv.txt 2>&1
KS=768,3072 timeout 300 python tools/gemm256_ksweep.py 20480 3072 0 12 > $OUT/ksweep_m20480.txt 2>&1
timeout 300 python tools/gemm256_phase_prof.py > $OUT/phase_prof.txt 2>&1
timeout 300 python tools/mfma_calib.py 4000 > $OUT/mfma_calib.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
prof() { name=$1; pmc=$2; shift 2; timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d $OUT/$name -o p --output-format csv -- "$@" > $OUT/$name.log 2>&1; \
         f=$(find $OUT/$name -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && python $REPO/tools/pmc_calib_summary.py $f 5 > $OUT/$name.md; rm -rf $OUT/$name; }
prof pmc1_calib "$P1" python $REPO/tools/mfma_calib.py 2000
prof pmc1_gemm "$P1" env VARS=0,12,4 KS=768,3072 python $REPO/tools/gemm256_variants_run.py
prof pmc2_gemm "$P2" env VARS=0,12 KS=768,3072 python $REPO/tools/gemm256_variants_run.py
cd $REPO
# in-step A/B on the same box: shipped schedule vs free-running form for every gemm256 launch (development library both times)
for i in 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline > $OUT/bench_8phase_$i.json 2> $OUT/bench_8phase_$i.err
  HCM_GEMM256_FREE=1 timeout 300 python bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline > $OUT/bench_free_$i.json 2> $OUT/bench_free_$i.err
done
ls -la $OUT
