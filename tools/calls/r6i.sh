mkdir -p gpurun_out/r6i
for k in NONE HCM_L3_SAFE HCM_NO_DEPTH_L3; do
  for i in 1 2; do
  r=$(env HCM_DEV_LIB=1 $k=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "passed|failed")
  echo "$k=1: $r" >> gpurun_out/r6i/knobs.txt
  done
done
cat gpurun_out/r6i/knobs.txt
