mkdir -p gpurun_out/r6d
python tools/stem_determinism.py 128 120 > gpurun_out/r6d/stem_det.txt 2>&1
for i in 1 2 3; do HCM_DEV_LIB=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | tail -n 2; done > gpurun_out/r6d/cfg4_default.txt
for i in 1 2 3; do HCM_DEV_LIB=1 HCM_NO_STEM_FUSE=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | tail -n 2; done > gpurun_out/r6d/cfg4_nofuse.txt
for i in 1 2 3; do HCM_DEV_LIB=1 HCM_NO_STEM_RED=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | tail -n 2; done > gpurun_out/r6d/cfg4_nored.txt
cat gpurun_out/r6d/stem_det.txt gpurun_out/r6d/cfg4_*.txt
