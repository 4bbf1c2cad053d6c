mkdir -p gpurun_out/r6j
for k in 1 2 3; do
  for i in 1 2; do
  r=$(env HCM_DEV_LIB=1 HCM_L3_SAFE=$k python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "passed|failed")
  echo "HCM_L3_SAFE=$k: $r" >> gpurun_out/r6j/knobs.txt
  done
done
cat gpurun_out/r6j/knobs.txt
