mkdir -p gpurun_out/r6f
python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "passed|failed" > gpurun_out/r6f/ship_on.txt
HCM_DEV_LIB=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "passed|failed" > gpurun_out/r6f/dev_fast.txt
HCM_DEV_LIB=1 HCM_NO_STEM_FUSE=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "passed|failed" > gpurun_out/r6f/dev_fast_nofuse.txt
head gpurun_out/r6f/*.txt
