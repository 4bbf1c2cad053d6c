cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "skinny" 2>&1 | tail -5 > gpurun_out/t_skinny.log
HCM_DEV_LIB=1 timeout 300 python tools/skinny_bench.py > gpurun_out/skinny_bench_w1.md 2>&1
HCM_DEV_LIB=1 HCM_SKINNY_WAVES=2 timeout 300 python tools/skinny_bench.py > gpurun_out/skinny_bench_w2.md 2>&1
cat gpurun_out/t_skinny.log gpurun_out/skinny_bench_w1.md gpurun_out/skinny_bench_w2.md
