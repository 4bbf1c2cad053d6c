python -m pytest tests/test_ops_gpu.py -q -k "bottleneck or bneck or fold or down" 2>&1 | tail -2
python -m pytest tests/test_fusion_toggles_gpu.py -q -x -k "rgb_trunk" 2>&1 | tail -2
for B in 8 128; do python tools/bneck_bench.py $B 64 64 1 2>/dev/null | grep "fused incl"; done
export HCM_DEV_LIB=1 HCM_IGEMM_PROF=1; for B in 8 128; do python tools/bneck_prof.py $B 64 2>/dev/null; done
