python -m pytest tests/test_ops_gpu.py -q -k "bottleneck or bneck or fold or down" 2>&1 | tail -2
python -m pytest tests/test_fusion_toggles_gpu.py -q -x -k "rgb_trunk" 2>&1 | tail -2
bash tools/profile_bench.sh r3d 1 ktonly > /dev/null 2>&1
grep "bneck" gpurun_out/prof_r3d_cfg1/r3d_kernel_trace_bench.md | cut -c1-140
