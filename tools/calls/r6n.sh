mkdir -p gpurun_out/r6ev
bash tools/profile_bench.sh r6 1 > gpurun_out/r6ev/profile_cfg1.log 2>&1
cp gpurun_out/prof_r6_cfg1/r6_*.md gpurun_out/prof_r6_cfg1/pmc_traffic*.json gpurun_out/r6ev/ 2>/dev/null
bash tools/chain_listing.sh r6ev 64 > /dev/null 2>&1
timeout 400 python tools/chain_times.py 64 > gpurun_out/r6ev/chain_times_b64.txt 2>&1
HCM_DEV_LIB=1 timeout 600 python tools/step_marks.py 64 > gpurun_out/r6ev/marks_b64.txt 2>&1
ls gpurun_out/r6ev; cat gpurun_out/r6ev/chain_times_b64.txt
