mkdir -p gpurun_out/r6e
for i in 1 2; do python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "^E  |passed|failed|Error" | head -8; done > gpurun_out/r6e/cfg4_ship.txt
HCM_DEV_LIB=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "^E  |passed|failed|Error" | head -12 > gpurun_out/r6e/cfg4_dev.txt
HCM_DEV_LIB=1 HCM_NO_STEM_FUSE=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "^E  |passed|failed|Error" | head -12 > gpurun_out/r6e/cfg4_nofuse.txt
cat gpurun_out/r6e/*.txt
