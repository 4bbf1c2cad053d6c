python -m pytest tests/test_ops_gpu.py -q -k "bottleneck or bneck or fold or down" 2>&1 | tail -2
python -m pytest tests/test_fusion_toggles_gpu.py -q -x -k "rgb_trunk" 2>&1 | tail -2
python tools/bneck_bench.py 2>/dev/null | grep "fused incl"
export HCM_DEV_LIB=1 HCM_IGEMM_PROF=1; for B in 128; do python tools/bneck_prof.py $B 32 128 2>/dev/null; done
